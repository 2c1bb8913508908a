#!/bin/bash
# usage (on the GPU box): bash tools/ab.sh TAG "ENV=.. ENV=.." [bench args]  -> one line: value, ms/step, kernel ms
TAG=$1; ENVS=$2; shift 2
env $ENVS timeout -k 10 200 python bench.py --one-rule --steps 20 --warmup 5 --no-cpu-baseline --no-k-cap --no-converged "$@" > gpurun_out/ab_$TAG.json 2> gpurun_out/ab_$TAG.err || { echo "$TAG FAILED"; tail -5 gpurun_out/ab_$TAG.err; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/ab_$TAG.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$TAG: %.1fM photons/s  %.3f ms/pass  kernel %.3f ms  frac %.3f  trace %.2f ms" % (d["value"] / 1e6, d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["roofline_trace"]["avg_launch_ms"]))
PY
