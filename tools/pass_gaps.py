#!/usr/bin/env python3
"""Per gather launch of a rocprofv3 --kernel-trace rocpd database: duration and the gap since the previous gather ended.
    pass_gaps.py RESULTS.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = [r for r in db.execute("select name, start, end from kernels order by start") if "k_gather_tiles" in r[0] or "k_deposit_tiles" in r[0]]
prev = None
print("launch  duration_us  gap_before_us")
for i, (name, s, e) in enumerate(rows):
    print("%5d  %10.1f  %10.1f" % (i, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
tot = (rows[-1][2] - rows[0][1]) / 1e3
busy = sum(e - s for _, s, e in rows) / 1e3
print("total %.1f us, inside the gather kernel %.1f us (%.1f %%)" % (tot, busy, 100.0 * busy / tot))
