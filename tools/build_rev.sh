#!/bin/bash
# Build the library of another git revision into ab_build/lib_<name>.so (same-box A/B timing: tools/gpu.sh <tag> ab:<name>).
#   usage: tools/build_rev.sh <rev> [name]
set -e
REV=$1; NAME=${2:-$1}
mkdir -p ab_build/src_$NAME
git archive $REV bde2vid_amd/csrc include Makefile | tar -x -C ab_build/src_$NAME
(cd ab_build/src_$NAME && make -s all && cp bde2vid_amd/libbde2vid.so ../lib_$NAME.so)
rm -rf ab_build/src_$NAME
echo built ab_build/lib_$NAME.so
