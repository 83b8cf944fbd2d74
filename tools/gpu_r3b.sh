#!/bin/bash
# full GPU suite + default bench + HD full-size verification
set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tee $O/r3b_pytest.txt | tail -8 || exit 1
timeout -k 10 400 python bench.py 2>$O/r3b_bench.err | tee $O/r3b_bench.json | cut -c1-300 || { tail -20 $O/r3b_bench.err; exit 1; }
timeout -k 10 500 python bench.py --height 720 --width 1280 --seq-len 64 --pipeline 1 --steps 2 --warmup 1 --no-cpu-baseline 2>$O/r3b_hd.err | tee $O/r3b_hd.json | cut -c1-300 || { tail -20 $O/r3b_hd.err; exit 1; }
