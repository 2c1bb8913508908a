run() { timeout -k 10 150 python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1),'Mph/s', round(d['ms_per_step'],3),'ms/step kernel', round(d['roofline']['avg_launch_ms'],3))" || exit 1; }
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread || exit 1
echo rule1 40; run --rule 1 --steps 40 --warmup 2
echo rule0 40; run --rule 0 --steps 40 --warmup 2
echo rule0 20/5; run --rule 0 --steps 20 --warmup 5
echo rule1 20/5; run --rule 1 --steps 20 --warmup 5
echo rule1 ahead2; SPPM_AHEAD=2 run --rule 1 --steps 40 --warmup 2
echo rule1 ahead3; SPPM_AHEAD=3 run --rule 1 --steps 40 --warmup 2
echo rule0 256; run --rule 0 --steps 256 --warmup 2
echo rule1 256; run --rule 1 --steps 256 --warmup 2
