#!/usr/bin/env python3
"""Kernel timeline between two consecutive k_gather_tiles launches of a rocprofv3 --kernel-trace rocpd database.

    timeline_db.py RESULTS.db [index of the gather counted from the end, default 100] [shortest kernel shown in us, default 10]
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
floor_us = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
rows = list(db.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
g = [i for i, r in enumerate(rows) if "k_gather_tiles" in r[0] or "k_deposit_tiles" in r[0]]
a, b = g[-1 - k], g[-k]
t0 = rows[a][1]
print("gather to gather: %.1f us" % ((rows[b][1] - rows[a][1]) / 1e3))
lo = a
while lo > 0 and rows[lo - 1][2] > t0:
    lo -= 1
for name, s, e, st, q in rows[max(0, a - 40):b + 1]:
    if e < t0 or (e - s) / 1e3 < floor_us:
        continue
    print("%9.1f us  +%8.1f us  stream %-3s queue %-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, st, q, name.split("(")[0].replace("void ", "")[:48]))
