#!/bin/bash
# Diagnostic (GPU box): marginal cost of each stage in the pipelined bench, by skipping it (results invalid).
# usage: tools/whatif.sh [bench args]
set -o pipefail
mkdir -p gpurun_out
for mask in 0 1 2 4 8 16 3 0; do
  BDE_TUNING="debug_skip=$mask" timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skip=$mask', 'ms/step', round(d['ms_per_step'],2), 'fps', round(d['value'],1))" \
    | tee -a gpurun_out/whatif.log || exit 1
done
