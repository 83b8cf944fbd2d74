#!/bin/bash
# GPU box (through gpurun): tests, bench, rocprofv3 kernel-trace summaries.   usage: tools/gpu_round.sh <tag> [notests]
set -o pipefail
TAG=${1:-r2}
R=$PWD
mkdir -p $R/gpurun_out
if [ "$2" != "notests" ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tee $R/gpurun_out/${TAG}_pytest.txt | tail -15 || exit 1
fi
timeout -k 10 500 python bench.py 2>$R/gpurun_out/${TAG}_bench.err | tee $R/gpurun_out/${TAG}_bench.json | cut -c1-300 || { tail -20 $R/gpurun_out/${TAG}_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
for mode in default pipeline1; do
  ARGS="--steps 6 --warmup 2 --no-cpu-baseline"; [ $mode = pipeline1 ] && ARGS="$ARGS --pipeline 1"
  rm -rf $R/gpurun_out/prof_${TAG}_$mode
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$mode -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_$mode.log 2>&1 || { tail -5 $R/gpurun_out/prof_${TAG}_$mode.log; exit 1; }
  (cd $R && NF=17; [ $mode = pipeline1 ] && NF=13; python tools/prof_summary.py gpurun_out/prof_${TAG}_$mode $NF > gpurun_out/${TAG}_${mode}_summary.txt; cp $(find gpurun_out/prof_${TAG}_$mode -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_${mode}_kernel_stats.csv)
done
cd $R
head -32 gpurun_out/${TAG}_pipeline1_summary.txt
