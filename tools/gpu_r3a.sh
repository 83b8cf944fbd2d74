#!/bin/bash
# round-3 first GPU pass: parity tests (incl. full-size VGA), default bench, full-size bench verification, batch-vs-pipeline A/B
set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "not 720x1280_T64" 2>&1 | tee $O/r3a_pytest.txt | tail -15 || exit 1
timeout -k 10 400 python bench.py 2>$O/r3a_bench.err | tee $O/r3a_bench.json | cut -c1-400 || { tail -20 $O/r3a_bench.err; exit 1; }
timeout -k 10 400 python bench.py --height 480 --width 640 --seq-len 32 --batch 4 --pipeline 1 --steps 3 --warmup 1 --no-cpu-baseline 2>$O/r3a_vga.err | tee $O/r3a_vga.json | cut -c1-400 || { tail -20 $O/r3a_vga.err; exit 1; }
timeout -k 10 300 python bench.py --batch 3 --pipeline 1 --steps 10 --warmup 2 --no-cpu-baseline 2>$O/r3a_b3p1.err | tee $O/r3a_b3p1.json | cut -c1-300 || { tail -20 $O/r3a_b3p1.err; exit 1; }
