timeout -k 10 400 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -2
tools/kstat.sh auto_dec dec "conv_"
for i in 1 2; do timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],2), 'fps', round(d['value'],1))"; done
