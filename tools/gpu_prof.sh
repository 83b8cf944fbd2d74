#!/bin/bash
# Run on the GPU box (through gpurun): tests, bench, rocprofv3 kernel trace of the bench.
# usage: tools/gpu_prof.sh <tag> [bench args]
set -o pipefail
TAG=$1; shift
R=$PWD
mkdir -p $R/gpurun_out
timeout -k 10 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tee $R/gpurun_out/bench_$TAG.json | cut -c1-330 || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 1
cd $R && python tools/prof_summary.py gpurun_out/prof_$TAG 5 | tee gpurun_out/prof_${TAG}_summary.txt
python - <<PY
import json
d=json.load(open('gpurun_out/bench_$TAG.json'))
print('fps', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'lstm0 avg us', round(d['roofline']['avg_us'],1), 'frac', round(d['roofline']['frac'],3))
PY
