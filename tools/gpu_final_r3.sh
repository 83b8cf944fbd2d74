#!/bin/bash
# GPU box (through gpurun): the artefacts committed under profiles/ for round 3.   usage: tools/gpu_final_r3.sh <tag>
set -o pipefail
TAG=${1:-r3}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 500 python bench.py 2>$O/${TAG}_bench.err | tee $O/${TAG}_bench.json | cut -c1-200 || { tail -20 $O/${TAG}_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
for mode in default pipeline1; do
  ARGS="--steps 6 --warmup 2 --no-cpu-baseline"; [ $mode = pipeline1 ] && ARGS="$ARGS --pipeline 1"
  rm -rf $O/prof_${TAG}_$mode
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_$mode -- python3 $R/bench.py $ARGS > $O/prof_${TAG}_$mode.log 2>&1 || { tail -5 $O/prof_${TAG}_$mode.log; exit 1; }
  (cd $R && python tools/prof_summary.py gpurun_out/prof_${TAG}_$mode $([ $mode = pipeline1 ] && echo 21 || echo 25) > gpurun_out/${TAG}_${mode}_summary.txt; cp $(find gpurun_out/prof_${TAG}_$mode -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_${mode}_kernel_stats.csv)
done
cd $R
tools/gpu_pmc_all.sh $TAG > $O/${TAG}_pmc_run.log 2>&1 || { tail -5 $O/${TAG}_pmc_run.log; exit 1; }
head -24 $O/${TAG}_pipeline1_summary.txt
