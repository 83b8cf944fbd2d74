#!/bin/bash
# GPU box: bench lines of BASELINE configs 3 (VGA, T = 32, B = 4) and 5 (HD, T = 64) at full size, each verified against its
# reference fixture.   usage: tools/gpu_configs.sh <tag> [pipeline]
set -o pipefail
TAG=${1:-r3}; PL=${2:-2}
O=gpurun_out; mkdir -p $O
timeout -k 10 500 python bench.py --height 480 --width 640 --seq-len 32 --batch 4 --pipeline $PL --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_config3.json 2>$O/${TAG}_config3.err || { tail -5 $O/${TAG}_config3.err; exit 1; }
timeout -k 10 500 python bench.py --height 720 --width 1280 --seq-len 64 --pipeline $PL --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_config5.json 2>$O/${TAG}_config5.err || { tail -5 $O/${TAG}_config5.err; exit 1; }
for c in 3 5; do python - <<P
import json
d = json.load(open('$O/${TAG}_bench_config$c.json'))
print('config $c:', round(d['value'], 1), 'frames/s, single stream', round(d['single_stream']['value'], 1), 'verified', d['verified'], d['verification'].get('max_abs_err'))
print('   ', {n: round(v['avg_us'], 1) for n, v in d['roofline']['kernels'].items()})
P
done
