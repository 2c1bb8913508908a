# usage (on the GPU box): bash tools/pmc_passes.sh TAG <bench.py arguments>
# hardware counters of one bench.py run, one rocprofv3 --pmc run per group (a run holds at most 8 SQ counters, 4 TCC);
# -> gpurun_out/pmc_TAG/sq_counters.json (per kernel, per launch averages)
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $grp -d $O/p$i -o p -- python3 $R/bench.py --no-cpu-baseline --no-k-cap "$@" > $O/bench_p$i.json 2>$O/p$i.err
  echo "pass $i done"
done
cd $R
python tools/rocpd_summary.py sq $O/sq_counters.json $(find $O -name 'p_results.db' | sort)
find $O -name '*.db' -delete
