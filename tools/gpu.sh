#!/bin/bash
# GPU box (through gpurun): ONE parameterised script for a round's measurements.
#   usage: tools/gpu.sh <tag> <step> [<step> ...]
# Steps run in the order given and stop at the first failure (a GPU step that failed or timed out is never followed by another):
#   tests            pytest -m gpu                                            -> gpurun_out/<tag>_pytest.txt
#   tests:<expr>     pytest -m gpu -k <expr>
#   bench            python bench.py (default command)                        -> gpurun_out/<tag>_bench.json
#   bench:<name>:<args...>   bench.py with extra arguments (',' separates them)  -> gpurun_out/<tag>_bench_<name>.json
#   ab:<lib>         bench.py --no-cpu-baseline with BDE_LIB_PATH=ab_build/lib_<lib>.so (same box A/B; boxes differ by up to 25 %)
#   tune:<name>:<k=v,...>    bench.py --no-cpu-baseline with BDE_TUNING=<k=v,...>
#   prof             rocprofv3 --kernel-trace --stats of the default and the --pipeline 1 command
#                                                                             -> gpurun_out/<tag>_{default,pipeline1}_{summary.txt,kernel_stats.csv}
#   pmc              FETCH_SIZE / WRITE_SIZE / SQ_* counter passes (each its own run, --kernel-trace only) -> profiles/<tag>_pmc.json
#   configs          bench lines of BASELINE configs 3 and 5 at full size     -> gpurun_out/<tag>_bench_config{3,5}.json
#   py:<script>:<args...>    python tools/<script>.py args                    -> gpurun_out/<tag>_<script>.txt
#   pyv:<lib>:<script>:<args...>   the same with BDE_LIB_PATH=ab_build/lib_<lib>.so
set -o pipefail
TAG=$1; shift
R=$PWD; O=$R/gpurun_out; mkdir -p $O
FRAMES_DEFAULT=23; FRAMES_P1=21          # forwards of `bench.py --steps 6 --warmup 2 --no-strict` with 2 / 1 sequences in flight (prof_summary.py)

run_bench() {   # name, env assignments..., then "--" and bench arguments
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 560 python bench.py "$@" 2>$O/${TAG}_bench${name:+_$name}.err | tee $O/${TAG}_bench${name:+_$name}.json | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$TAG bench ${name:-default}]', round(d['value'],1), 'fps', d['ms_per_step'], 'ms/step; single', d.get('single_stream_fps', (d.get('single_stream') or {}).get('value')), 'verified', d.get('verified'))" \
    || { tail -20 $O/${TAG}_bench${name:+_$name}.err; return 1; }
}

for step in "$@"; do
  echo "=== [$TAG] $step"
  case $step in
    tests)   timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tee $O/${TAG}_pytest.txt | tail -15 || exit 1 ;;
    tests:*) timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "${step#tests:}" 2>&1 | tee $O/${TAG}_pytest_k.txt | tail -15 || exit 1 ;;
    bench)   run_bench "" -- || exit 1 ;;
    bench:*) IFS=: read -r _ name args <<< "$step"; run_bench "$name" -- ${args//,/ } || exit 1 ;;
    ab:*)    lib=${step#ab:}; run_bench "ab_$lib" BDE_LIB_PATH=$R/ab_build/lib_$lib.so BDE_LIB_ANY_ABI=1 -- --no-cpu-baseline --no-strict || exit 1 ;;
    tune:*)  IFS=: read -r _ name kv <<< "$step"; run_bench "$name" BDE_TUNING=$kv -- --no-cpu-baseline --no-strict || exit 1 ;;
    prof)
      cd /tmp && export TMPDIR=/tmp
      for mode in default pipeline1; do
        ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-strict"; NF=$FRAMES_DEFAULT
        [ $mode = pipeline1 ] && { ARGS="$ARGS --pipeline 1"; NF=$FRAMES_P1; }
        rm -rf $O/prof_${TAG}_$mode
        timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_$mode -- python3 $R/bench.py $ARGS > $O/prof_${TAG}_$mode.log 2>&1 \
          || { tail -5 $O/prof_${TAG}_$mode.log; exit 1; }
        (cd $R && python tools/prof_summary.py gpurun_out/prof_${TAG}_$mode $NF > gpurun_out/${TAG}_${mode}_summary.txt &&
          cp $(find gpurun_out/prof_${TAG}_$mode -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_${mode}_kernel_stats.csv) || exit 1
      done
      cd $R; head -30 $O/${TAG}_pipeline1_summary.txt ;;
    pmc)
      cd /tmp && export TMPDIR=/tmp
      ARGS="--pipeline 1 --steps 2 --warmup 1 --no-cpu-baseline --no-strict"
      for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
        d=pmc_${TAG}_${c%% *}; [ "${c%% *}" = SQ_WAVE_CYCLES ] && d=pmc_${TAG}_SQ
        rm -rf $O/$d; echo "[pmc] pass ${c%% *}"
        timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$d -- python3 $R/bench.py $ARGS > $O/$d.log 2>&1 || { tail -5 $O/$d.log; exit 1; }
      done
      cd $R && python tools/pmc_collect.py $TAG gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE gpurun_out/pmc_${TAG}_SQ | tee gpurun_out/${TAG}_pmc.txt || exit 1
      cp profiles/${TAG}_pmc.json gpurun_out/${TAG}_pmc.json ;;
    configs)
      run_bench config3 -- --height 480 --width 640 --seq-len 32 --batch 4 --steps 4 --warmup 1 --pipeline 2 --no-cpu-baseline --no-strict || exit 1
      run_bench config5 -- --height 720 --width 1280 --seq-len 64 --steps 3 --warmup 1 --pipeline 2 --no-cpu-baseline --no-strict || exit 1 ;;
    py:*)    IFS=: read -r _ script args <<< "$step"
             timeout -k 10 500 python tools/$script.py ${args//,/ } 2>&1 | tee $O/${TAG}_$script.txt | tail -40 || exit 1 ;;
    pyv:*)   IFS=: read -r _ lib script args <<< "$step"
             BDE_LIB_PATH=$R/ab_build/lib_$lib.so BDE_LIB_ANY_ABI=1 timeout -k 10 500 python tools/$script.py ${args//,/ } 2>&1 | tee $O/${TAG}_${script}_$lib.txt | tail -40 || exit 1 ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
