timeout -k 10 400 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -3
for st in lstm0 lstm1 lstm2; do
LSTM_GLDS=0 tools/kstat.sh reg_$st $st "lstm16"
tools/kstat.sh glds_$st $st "lstm16"
done
