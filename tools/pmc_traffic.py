#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one run each) into profiles/r01_pmc_traffic.json.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write WIDTH HEIGHT PHOTONS STEPS

FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived counters); FETCH_SIZE is doubled, the gfx950
correction of /opt/skills/guides/MI355X_MICROARCH.md (HBM section)."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    tot, calls = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[k] += float(r["Counter_Value"])
            calls[k] += 1
    return tot, calls


fetch, fc = per_kernel(sys.argv[1], "FETCH_SIZE")
write, wc = per_kernel(sys.argv[2], "WRITE_SIZE")
w, h, m, steps = [int(x) for x in sys.argv[3:7]]
rows = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, 0) + write.get(k, 0))):
    rows[k] = {"launches": fc.get(k, wc.get(k)), "fetch_bytes_per_launch": 2 * 1024 * fetch.get(k, 0) / max(1, fc.get(k, 0)),
               "write_bytes_per_launch": 1024 * write.get(k, 0) / max(1, wc.get(k, 0))}
g = next(v for k, v in rows.items() if k.startswith("k_gather_tiles"))
out = {"width": w, "height": h, "photons": m, "steps": steps,
       "gather_bytes_per_launch": g["fetch_bytes_per_launch"] + g["write_bytes_per_launch"],
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py; KiB -> bytes; FETCH_SIZE x2 (gfx950)",
       "kernels": rows}
json.dump(out, open("profiles/r01_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("gather_bytes_per_launch",)}))
