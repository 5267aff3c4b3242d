python -m pytest tests/test_gpu_parity.py tests/test_gpu_blocks.py -m gpu -q -x -k "gauss or gradient or nan or finite" 2>&1 | grep -E "passed|failed|Error|assert" | head
python tools/gauss_edge_sweep.py 2>&1 | tail -2
python tools/nan_sea_time.py 8192
