#!/bin/bash
# Build the library of another git revision into ab_build/lib_<rev>.so (for same-box A/B timing:
# BDE_LIB_PATH=ab_build/lib_<rev>.so python tools/microbench.py ...).
set -e
REV=$1
mkdir -p ab_build/src_$REV
git archive $REV bde2vid_amd/csrc include | tar -x -C ab_build/src_$REV
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden -DBDE_BUILD \
    -o ab_build/lib_$REV.so ab_build/src_$REV/bde2vid_amd/csrc/bde_api.hip
rm -rf ab_build/src_$REV
echo built ab_build/lib_$REV.so
