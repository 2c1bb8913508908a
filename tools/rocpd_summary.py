#!/usr/bin/env python3
"""Summaries of rocprofv3's rocpd (.db) output, the format this ROCm writes by default.

    rocpd_summary.py stats  RESULTS.db OUT.csv                      kernel statistics (what --stats prints)
    rocpd_summary.py pmc    FETCH.db WRITE.db W H PHOTONS STEPS     -> profiles/r01_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived counters); FETCH_SIZE is doubled, the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (HBM section).  Each counter comes from a run of its own."""
import collections
import csv
import json
import sqlite3
import statistics
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")


def stats(db_path, out_csv):
    db = sqlite3.connect(db_path)
    d = collections.defaultdict(list)
    for name, dur in db.execute("select name, duration from kernels"):
        d[name].append(dur)
    total = sum(sum(v) for v in d.values())
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            n = name if len(name) < 200 else short(name)[:160] + "<...>"
            w.writerow([n, len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / total, 2), min(v), max(v),
                        round(statistics.pstdev(v), 3)])


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    tot, calls = collections.defaultdict(float), collections.Counter()
    for name, val in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        tot[short(name)] += val
        calls[short(name)] += 1
    return tot, calls


def pmc(fetch_db, write_db, w, h, m, steps):
    fetch, fc = per_kernel(fetch_db, "FETCH_SIZE")
    write, wc = per_kernel(write_db, "WRITE_SIZE")
    rows = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, 0) + write.get(k, 0))):
        if len(k) > 120:
            continue
        rows[k] = {"launches": fc.get(k, wc.get(k)), "fetch_bytes_per_launch": 2 * 1024 * fetch.get(k, 0) / max(1, fc.get(k, 0)),
                   "write_bytes_per_launch": 1024 * write.get(k, 0) / max(1, wc.get(k, 0))}
    g = next(v for k, v in rows.items() if k.startswith("k_gather_tiles"))
    out = {"width": w, "height": h, "photons": m, "steps": steps,
           "gather_bytes_per_launch": g["fetch_bytes_per_launch"] + g["write_bytes_per_launch"],
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py; KiB -> bytes; FETCH_SIZE x2 (gfx950)",
           "kernels": rows}
    json.dump(out, open("profiles/r01_pmc_traffic.json", "w"), indent=1)
    print(json.dumps({"gather_bytes_per_launch": out["gather_bytes_per_launch"], "launches": g["launches"]}))


def sq(out_json, dbs):
    """per kernel: launches and the per-launch average of every counter found in the given --pmc databases"""
    out = {}
    for path in dbs:
        db = sqlite3.connect(path)
        tot, calls = collections.defaultdict(float), collections.defaultdict(set)
        for name, counter, val, disp in db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
            k = short(name)
            if len(k) > 120:
                continue
            tot[(k, counter)] += val
            calls[(k, counter)].add(disp)
        for (k, counter), v in tot.items():
            n = max(1, len(calls[(k, counter)]))
            out.setdefault(k, {"launches": n})[counter] = v / n
    for k, d in out.items():   # derived figures the guide names
        if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = 2 * 1024 * d.get("FETCH_SIZE", 0.0) + 1024 * d.get("WRITE_SIZE", 0.0)
        if d.get("SQ_BUSY_CYCLES") and d.get("SQ_ACTIVE_INST_VALU"):
            d["valu_busy_share_of_sq_busy_cycles"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_BUSY_CYCLES"]
    keep = {k: v for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0))}
    json.dump({"method": "rocprofv3 --pmc, one run per group of counters (tools/pmc_passes.sh); values are per-launch averages summed over "
                         "all XCDs / shader engines; FETCH_SIZE and WRITE_SIZE in KiB, hbm_bytes_per_launch = 2 x FETCH_SIZE (gfx950) + WRITE_SIZE in bytes",
               "kernels": keep}, open(out_json, "w"), indent=1)
    for k in list(keep)[:6]:
        print(k, json.dumps(keep[k]))


def pmcfile(sq_json, kernel_prefix, out_json, w, h, m, rule, run):
    """the compact per-kernel file bench.py reads (profiles/r02_pmc_rule<R>.json) from a tools/pmc_passes.sh summary.
    Shares: persistent wavefronts live about as long as the kernel, so L = SQ_WAVE_CYCLES / SQ_WAVES (units of four
    cycles, like SQ_ACTIVE_INST_*) is the time every SIMD was occupied; valu_active_share = SQ_ACTIVE_INST_VALU / 1024
    SIMDs / L; lds_active_share = SQ_LDS_IDX_ACTIVE (cycles) / 256 CUs / 4 L."""
    k = json.load(open(sq_json))["kernels"]
    name = next(n for n in k if n.startswith(kernel_prefix))
    d = k[name]
    L = d["SQ_WAVE_CYCLES"] / d["SQ_WAVES"]
    out = {"width": w, "height": h, "photons": m, "rule": rule, "kernel": name, "run": run, "launches": d["launches"],
           "hbm_bytes_per_launch": d["hbm_bytes_per_launch"],
           "valu_active_share": d["SQ_ACTIVE_INST_VALU"] / 1024.0 / L,
           "lds_active_share": d["SQ_LDS_IDX_ACTIVE"] / 256.0 / (4.0 * L),
           "salu_insts_per_valu_inst": d["SQ_INSTS_SALU"] / d["SQ_INSTS_VALU"],
           "valu_insts_per_launch": d["SQ_INSTS_VALU"], "lds_insts_per_launch": d["SQ_INSTS_LDS"],
           "fp64_insts_per_launch": d["SQ_INSTS_VALU_ADD_F64"] + d["SQ_INSTS_VALU_MUL_F64"] + d["SQ_INSTS_VALU_FMA_F64"],
           "lds_bank_conflict_share": d["SQ_LDS_BANK_CONFLICT"] / max(1.0, d["SQ_LDS_IDX_ACTIVE"]),
           "wait_inst_any_share": d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"],
           "method": "rocprofv3 --pmc, one run per counter group (tools/pmc_passes.sh); per-launch averages; HBM bytes = 2 x FETCH_SIZE (gfx950) + WRITE_SIZE",
           "counters": d}
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps({x: out[x] for x in ("kernel", "hbm_bytes_per_launch", "valu_active_share", "lds_active_share", "wait_inst_any_share")}))


if __name__ == "__main__":
    if sys.argv[1] == "pmcfile":
        pmcfile(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), sys.argv[9])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "sq":
        sq(sys.argv[2], sys.argv[3:])
    else:
        pmc(sys.argv[2], sys.argv[3], *[int(x) for x in sys.argv[4:8]])
