#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> <stage> "<counters>"   (run through gpurun)
TAG=$1; STAGE=$2; CTRS=$3
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_$TAG
timeout -k 10 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/tools/microbench.py $STAGE 1 > $R/gpurun_out/pmc_$TAG.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/pmc_$TAG
