#!/bin/bash
# usage (on the GPU box): bash tools/bench_pair.sh [rule]   -- the bench at the driver's step count and over 256 passes, one rule, no CPU legs
R=${1:-1}
timeout -k 10 150 python bench.py --rule $R --steps 20 --warmup 5 --one-rule --no-cpu-baseline --no-k-cap > gpurun_out/b20.json 2>gpurun_out/b20.err || exit 1
timeout -k 10 250 python bench.py --rule $R --steps 256 --warmup 2 --one-rule --no-cpu-baseline --no-k-cap > gpurun_out/b256.json 2>gpurun_out/b256.err || exit 1
python3 - <<'PY'
import json
for f in ("b20", "b256"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, round(d["value"] / 1e6, 1), round(d["ms_per_step"], 3), round(r["avg_launch_ms"], 3), round(d["roofline_trace"]["avg_launch_ms"], 3))
PY
