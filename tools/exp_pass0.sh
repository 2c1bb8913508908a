#!/bin/bash
# usage (GPU box): bash tools/exp_pass0.sh <lib.so> <tag>  -- deposit kernel durations of a 3-pass 1080p render with another build of the library
export SPPM_HIP_LIB=$1
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/exp_$2 -o x -- python3 $GRAFT_REPO_ROOT/tools/dep_stamps.py 3 > /dev/null 2>&1 || exit 1
python3 - "$GRAFT_REPO_ROOT/gpurun_out/exp_$2" "$2" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in csv.DictReader(open(f)) if "k_deposit_tiles" in r["Kernel_Name"]]
print(sys.argv[2], " ".join("%.0f" % us for _, us in sorted(d)), "us")
PY
