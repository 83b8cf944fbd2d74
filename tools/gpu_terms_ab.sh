#!/bin/bash
# GPU box: GPU tests in the default split format, then bench.py in both formats on the same box.   usage: tools/gpu_terms_ab.sh <tag> [notests]
set -o pipefail
TAG=${1:-ab}
O=gpurun_out; mkdir -p $O
if [ "$2" != "notests" ]; then
  timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/${TAG}_pytest.txt 2>&1; rc=$?
  tail -25 $O/${TAG}_pytest.txt
  [ $rc = 0 ] || exit 1
fi
for t in 2 3; do
  BDE_TUNING=sb_terms=$t timeout -k 10 300 python bench.py --no-cpu-baseline 2>$O/${TAG}_t$t.err > $O/${TAG}_t$t.json || { tail -20 $O/${TAG}_t$t.err; exit 1; }
  python - <<P
import json
d = json.load(open('$O/${TAG}_t$t.json'))
k = d['roofline']['kernels']
print('terms $t: value', round(d['value'], 1), 'single', round(d['single_stream']['value'], 1), 'verified', d['verified'], d['verification']['max_abs_err'])
print('   ', {n: round(v['avg_us'], 1) for n, v in k.items()})
P
done
