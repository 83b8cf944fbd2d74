timeout -k 10 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
BDE_TUNING="kv_ride=0" timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('noride', 'ms/step', round(d['ms_per_step'],2), 'fps', round(d['value'],1))"
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ride', 'ms/step', round(d['ms_per_step'],2), 'fps', round(d['value'],1))"
done
BDE_TUNING="kv_ride=0" timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --pipeline 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('noride p1', 'ms/step', round(d['ms_per_step'],2), 'fps', round(d['value'],1))"
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --pipeline 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ride p1', 'ms/step', round(d['ms_per_step'],2), 'fps', round(d['value'],1))"
