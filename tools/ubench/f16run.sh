python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gauss or gradient" 2>&1 | grep -E "passed|failed|Error|assert" | head
python tools/grad_time.py 1.25 3.25 6.0
