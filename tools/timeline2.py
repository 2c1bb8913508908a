"""Kernel timeline between the last two k_gather_tiles launches of a rocprofv3 --kernel-trace csv (all queues)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
g = [i for i, r in enumerate(rows) if 'k_gather_tiles' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a, b = g[-1 - k], g[-k]
t0 = int(rows[a]['End_Timestamp'])
for r in rows[a:b + 1]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if d < 15: continue
    print('%9.1f us  +%8.1f us  q%s  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, d, r.get('Queue_Id', '?'), r['Kernel_Name'][:60]))
