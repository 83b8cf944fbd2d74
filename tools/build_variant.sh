#!/bin/bash
# Build ab_build/lib_<name>.so from the working tree with extra -D flags (A/B timing of experiment switches on one box:
# BDE_LIB_PATH=ab_build/lib_<name>.so python tools/win_bench.py).   usage: tools/build_variant.sh <name> [-DFOO=1 ...]
set -e
NAME=$1; shift
mkdir -p ab_build build
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -fvisibility=hidden -DBDE_BUILD"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o ab_build/api_$NAME.o bde2vid_amd/csrc/bde_api.hip
[ -f build/conv_tu.o -a -f build/sb_tu.o ] || make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_build/lib_$NAME.so ab_build/api_$NAME.o build/conv_tu.o build/sb_tu.o
rm -f ab_build/api_$NAME.o
echo built ab_build/lib_$NAME.so
