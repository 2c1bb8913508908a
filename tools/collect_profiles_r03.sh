#!/bin/bash
# usage (here, after `gpurun -- bash tools/refresh_profiles_r03.sh prof|pmc|bench`): bash tools/collect_profiles_r03.sh -- copies the merged results from gpurun_out/ into profiles/r03_*
cd "$(dirname "$0")/.."
for r in 0 1; do
  cp gpurun_out/pmc_rule$r/pmc.json profiles/r03_pmc_rule$r.json
  cp gpurun_out/pmc_rule$r/sq_counters.json profiles/r03_sq_counters_rule$r.json
  cp gpurun_out/prof_rule$r/kernel_stats.csv profiles/r03_kernel_stats_rule$r.csv
  tail -1 gpurun_out/prof_rule$r/bench.json > profiles/r03_bench_profiled_rule$r.json
  cp gpurun_out/prof_rule$r/pass_gaps.txt profiles/r03_pass_gaps_rule$r.txt
  cp gpurun_out/prof_rule$r/timeline.txt profiles/r03_timeline_rule$r.txt
done
tail -1 gpurun_out/final3/bench.json > profiles/r03_bench.json
tail -1 gpurun_out/final3/bench_256.json > profiles/r03_bench_256.json
tail -1 gpurun_out/final3/bench_ppm.json > profiles/r03_bench_classic_ppm.json
tail -1 gpurun_out/final3/bench_10k.json > profiles/r03_bench_10k_spheres_4k.json
