timeout -k 10 400 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -3
for st in lstm0 lstm1 lstm2; do
BDE_LIB_PATH=$PWD/ab_build/lib_HEAD.so tools/kstat.sh old_$st $st "lstm16"
tools/kstat.sh new_$st $st "lstm16"
done
