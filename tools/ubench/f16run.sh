python -m pytest tests/test_gpu_parity.py tests/test_gpu_blocks.py tests/test_gpu_halo_loopback.py -m gpu -q -x -k "gauss or gradient or aspect or nan or config3 or config5 or blocks or loopback" 2>&1 | grep -E "passed|failed|Error|assert" | head
python tools/grad_time.py 8.0 10.0 11.75
