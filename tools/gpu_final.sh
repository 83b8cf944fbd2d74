#!/bin/bash
# GPU box (through gpurun): the artefacts committed under profiles/ for a round.
# usage: tools/gpu_final.sh <tag>      writes gpurun_out/<tag>_*
set -o pipefail
TAG=${1:-r1b}
R=$PWD
mkdir -p $R/gpurun_out
timeout -k 10 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -2 || exit 1
timeout -k 10 500 python bench.py 2>$R/gpurun_out/${TAG}_bench.err | tee $R/gpurun_out/${TAG}_bench.json | cut -c1-200 || exit 1
cd /tmp && export TMPDIR=/tmp
for mode in default pipeline1; do
  ARGS="--steps 6 --warmup 2 --no-cpu-baseline"; [ $mode = pipeline1 ] && ARGS="$ARGS --pipeline 1"
  rm -rf $R/gpurun_out/prof_${TAG}_$mode
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$mode -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_$mode.log 2>&1 || exit 1
  (cd $R && python tools/prof_summary.py gpurun_out/prof_${TAG}_$mode 13 > gpurun_out/${TAG}_${mode}_summary.txt; cp $(find gpurun_out/prof_${TAG}_$mode -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_${mode}_kernel_stats.csv)
done
cd $R
# HBM traffic of the recurrent step kernel: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, rocprofv3 PMC slots)
tools/gpu_pmc.sh ${TAG}_fetch lstm0 "FETCH_SIZE" > gpurun_out/${TAG}_pmc_fetch.txt 2>&1   # (FETCH_SIZE takes 3 of the 4 TCC slots: alone)
tools/gpu_pmc.sh ${TAG}_write lstm0 "WRITE_SIZE" > gpurun_out/${TAG}_pmc_write.txt 2>&1
tools/gpu_pmc.sh ${TAG}_sq lstm0 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" > gpurun_out/${TAG}_pmc_sq.txt 2>&1
grep -A4 "lstm16" gpurun_out/${TAG}_pmc_fetch.txt | head -8; grep -A2 "lstm16" gpurun_out/${TAG}_pmc_write.txt | head -4
cat gpurun_out/${TAG}_default_summary.txt | head -30
