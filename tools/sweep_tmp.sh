#!/bin/bash
# scratch: sweep SPPM_GATHER_BLOCKS, rule 0
for v in 1024 896 768 640; do
  export SPPM_GATHER_BLOCKS=$v
  timeout -k 10 150 python bench.py --rule 0 --steps 20 --warmup 5 --one-rule --no-cpu-baseline --no-k-cap > gpurun_out/b20_$v.json 2>gpurun_out/b20_$v.err || exit 1
  timeout -k 10 250 python bench.py --rule 0 --steps 256 --warmup 2 --one-rule --no-cpu-baseline --no-k-cap > gpurun_out/b256_$v.json 2>gpurun_out/b256_$v.err || exit 1
done
python3 - <<'PY'
import json
for v in ("1024", "896", "768", "640"):
    for f in ("b20", "b256"):
        d = json.loads(open("gpurun_out/%s_%s.json" % (f, v)).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f, v, round(d["value"] / 1e6, 1), round(d["ms_per_step"], 3), round(r["avg_launch_ms"], 3))
PY
