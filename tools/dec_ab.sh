#!/bin/bash
# decoder / head / encoder stage timings for a list of library variants on one box: tools/dec_ab.sh <tag> <lib> [<lib> ...]
TAG=$1; shift
O=gpurun_out; mkdir -p $O
for lib in "$@"; do
  for st in dec head; do
    echo "== $lib $st"
    BDE_LIB_PATH=$PWD/ab_build/lib_$lib.so BDE_LIB_ANY_ABI=1 timeout -k 10 200 python tools/microbench.py $st 5 2>&1 | grep -v amdgpu.ids
  done
done | tee $O/${TAG}_dec_ab.txt
