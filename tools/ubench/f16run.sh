python -m pytest tests/test_gpu_parity.py tests/test_gpu_blocks.py -m gpu -q -x -k "gauss or gradient or aspect or nan or config3 or config5 or blocks" 2>&1 | grep -E "passed|failed|Error|assert" | head
python tools/gauss_edge_sweep.py 2>&1 | tail -3
python tools/gauss_axes_time.py 30.25
python tools/grad_time.py 30.25
