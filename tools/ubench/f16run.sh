python -m pytest tests/test_gpu_parity.py tests/test_gpu_blocks.py -m gpu -q -x -k "gauss or gradient or aspect or nan or config3 or config5 or blocks" 2>&1 | grep -E "passed|failed|Error|assert" | head
python tools/grad_time.py 1.25 3.25 8.0 30.25
python tools/gauss_axes_time.py 3.25 8.0 30.25
