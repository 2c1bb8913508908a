# usage: bash tools/knob_sweep.sh   (on the GPU box) -- one bench.py line per setting: Mphotons/s, ms per pass, gather ms
# (results of the round: profiles/r01_knob_sweep.txt)
run() { echo -n "$* : "; env "$@" timeout -k 10 100 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3))"; }
run A=1
run SPPM_TRACE_BLOCKS=64
run SPPM_TRACE_BLOCKS=96
run SPPM_TRACE_BLOCKS=160
run SPPM_TRACE_BLOCKS=256
run SPPM_CELL_FACTOR=0.3
run SPPM_CELL_FACTOR=0.42
run SPPM_SPLIT_FACTOR=3
run SPPM_SPLIT_FACTOR=5.5
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=8
run A=2
