#!/bin/bash
# GPU box (through gpurun): HBM traffic counters of every kernel of the forward.  FETCH_SIZE (3 of the 4 TCC slots) and
# WRITE_SIZE in passes of their own, each with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots); the
# program after `--` is python3 itself.   usage: tools/gpu_pmc_all.sh <tag>   -> profiles/<tag>_pmc.json
set -o pipefail
TAG=${1:-r2}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--pipeline 1 --steps 2 --warmup 1 --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_${TAG}_$c
  echo "[pmc] pass $c"
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$c -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_$c.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${TAG}_$c.log; exit 1; }
done
rm -rf $R/gpurun_out/pmc_${TAG}_SQ
echo "[pmc] pass SQ"
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_SQ -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_SQ.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${TAG}_SQ.log; exit 1; }
cd $R && python tools/pmc_collect.py $TAG gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE gpurun_out/pmc_${TAG}_SQ | tee gpurun_out/${TAG}_pmc.txt
cp profiles/${TAG}_pmc.json gpurun_out/${TAG}_pmc.json
