set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/st -o final -- python3 $R/bench.py --no-cpu-baseline > $O/bench_profiled.json 2>$O/prof.err
echo profiled done
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/F -o f -- python3 $R/bench.py --no-cpu-baseline > $O/f.json 2>$O/f.err
echo fetch done
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/W -o w -- python3 $R/bench.py --no-cpu-baseline > $O/w.json 2>$O/w.err
echo write done
cd $R
python tools/rocpd_summary.py stats $(find $O/st -name 'final_results.db') $O/kernel_stats.csv
python tools/rocpd_summary.py pmc $(find $O/F -name 'f_results.db') $(find $O/W -name 'w_results.db') 1920 1080 1000000 256
cp profiles/r01_pmc_traffic.json $O/pmc_traffic.json
timeout -k 10 300 python bench.py > $O/bench.json 2>$O/bench.err
echo bench done
timeout -k 10 200 python bench.py --workload ppm --steps 64 > $O/bench_ppm.json 2>$O/ppm.err
echo ppm done
timeout -k 10 250 python bench.py --spheres 10000 --width 3840 --height 2160 --steps 32 --no-cpu-baseline > $O/bench_10k.json 2>$O/10k.err
echo 10k done
find $O -name '*.db' -delete
tail -c 600 $O/bench.json
