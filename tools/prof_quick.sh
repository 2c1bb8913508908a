# usage (on the GPU box): bash tools/prof_quick.sh TAG <bench.py arguments>
# kernel-trace statistics of one bench.py run -> gpurun_out/prof_TAG/{kernel_stats.csv,bench.json}
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $O/st -o run -- python3 $R/bench.py --no-cpu-baseline --no-k-cap "$@" > $O/bench.json 2>$O/prof.err
cd $R
python tools/rocpd_summary.py stats $(find $O/st -name 'run_results.db') $O/kernel_stats.csv
python tools/timeline_db.py $(find $O/st -name "run_results.db") 5 5 > $O/timeline.txt 2>&1 || true
python tools/pass_gaps.py $(find $O/st -name "run_results.db") > $O/pass_gaps.txt 2>&1 || true
find $O -name '*.db' -delete
head -30 $O/kernel_stats.csv
