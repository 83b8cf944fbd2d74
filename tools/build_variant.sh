#!/bin/bash
# Build ab_build/lib_<name>.so from the working tree with extra -D flags in ONE translation unit (A/B timing of experiment switches
# on one box: BDE_LIB_PATH=ab_build/lib_<name>.so python tools/win_bench.py).
#   usage: tools/build_variant.sh <name> <unit: bde_api | conv_tu | sb_tu> [-DFOO=1 ...]
set -e
NAME=$1; UNIT=$2; shift 2
mkdir -p ab_build build
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -fvisibility=hidden -DBDE_BUILD"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o ab_build/${UNIT}_$NAME.o bde2vid_amd/csrc/$UNIT.hip
OBJS=""
for u in bde_api conv_tu sb_tu; do
  if [ $u = $UNIT ]; then OBJS="$OBJS ab_build/${UNIT}_$NAME.o"; else [ -f build/$u.o ] || make -s; OBJS="$OBJS build/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_build/lib_$NAME.so $OBJS
rm -f ab_build/${UNIT}_$NAME.o
echo built ab_build/lib_$NAME.so
