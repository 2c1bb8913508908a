# usage (on the GPU box): bash tools/pmc_ifetch.sh -- instruction-cache and wait counters of the kernels of the driver's command (rule 1)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_ifetch; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $grp -d $O/p$i -o p -- python3 $R/bench.py --no-cpu-baseline --no-k-cap --rule 1 --one-rule --steps 20 --warmup 5 --no-converged > $O/bench_p$i.json 2>$O/p$i.err
  echo "pass $i done"
done
cd $R
python tools/rocpd_summary.py sq $O/sq.json $(find $O -name 'p_results.db' | sort)
find $O -name '*.db' -delete
python - <<PY
import json
d = json.load(open("$O/sq.json"))["kernels"]
for k, v in d.items():
    if any(n in k for n in ("deposit_tiles", "photon_trace", "eye_pass")): print(k[:40], {a: round(b, 1) for a, b in v.items()})
PY
