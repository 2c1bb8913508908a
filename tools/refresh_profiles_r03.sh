#!/bin/bash
# usage (on the GPU box, repository root): bash tools/refresh_profiles_r03.sh <part>   -- regenerates what profiles/r03_* holds
# part "prof": kernel statistics + timelines of the driver's command, both rules; "pmc": the counter passes, both rules;
# "bench": the unprofiled lines.  Results under gpurun_out/; copy them into profiles/ (see profiles/README.md).
set -e
R=$GRAFT_REPO_ROOT
RUN="python bench.py --rule %d --one-rule --steps 20 --warmup 5 --no-cpu-baseline --no-k-cap (warm-up passes included in the averages)"
case "$1" in
prof)
  for r in 1 0; do bash tools/prof_quick.sh rule$r --rule $r --one-rule --steps 20 --warmup 5 --no-converged > /dev/null; echo "prof rule $r done"; done ;;
pmc)
  for r in 1 0; do
    bash tools/pmc_passes.sh rule$r --rule $r --one-rule --steps 20 --warmup 5 --no-converged > /dev/null
    k=k_deposit_tiles; [ $r = 0 ] && k=k_gather_tiles
    python tools/rocpd_summary.py pmcfile gpurun_out/pmc_rule$r/sq_counters.json "void $k" gpurun_out/pmc_rule$r/pmc.json 1920 1080 1000000 $r "$(printf "$RUN" $r)" || \
    python tools/rocpd_summary.py pmcfile gpurun_out/pmc_rule$r/sq_counters.json "$k" gpurun_out/pmc_rule$r/pmc.json 1920 1080 1000000 $r "$(printf "$RUN" $r)"
    echo "pmc rule $r done"
  done ;;
bench)
  mkdir -p gpurun_out/final3
  timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final3/bench.json 2> gpurun_out/final3/bench.err; echo "bench done"
  timeout -k 10 400 python bench.py --steps 256 --warmup 2 --no-cpu-baseline --no-k-cap --no-converged > gpurun_out/final3/bench_256.json 2> gpurun_out/final3/bench_256.err; echo "256 done"
  timeout -k 10 300 python bench.py --workload ppm --steps 64 > gpurun_out/final3/bench_ppm.json 2> gpurun_out/final3/ppm.err; echo "ppm done"
  timeout -k 10 400 python bench.py --spheres 10000 --width 3840 --height 2160 --steps 32 --no-cpu-baseline --no-k-cap --no-converged > gpurun_out/final3/bench_10k.json 2> gpurun_out/final3/10k.err; echo "10k done" ;;
esac
