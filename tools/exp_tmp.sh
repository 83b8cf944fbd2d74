timeout -k 10 400 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -3
python tools/win_stamps.py 2>&1 | grep "phase\|block 0"
tools/kstat.sh wb attn0 "winblock"
