timeout -k 10 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
BDE_LIB_PATH=$PWD/ab_build/lib_HEAD.so timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', 'ms/step', round(d['ms_per_step'],2), 'fps', round(d['value'],1), 'lstm0 us', round(d['roofline']['avg_us'],1))"
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', 'ms/step', round(d['ms_per_step'],2), 'fps', round(d['value'],1), 'lstm0 us', round(d['roofline']['avg_us'],1))"
done
