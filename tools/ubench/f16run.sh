python -m pytest tests/test_gpu_parity.py tests/test_gpu_blocks.py -m gpu -q -x -k "gauss or gradient or nan or finite or widths or config3" 2>&1 | grep -E "passed|failed|Error|assert" | head
for a in 1 0; do TOPO_AMD_GAUSS_MFMA_ANY_WIDTH=$a python tools/odd_width_time.py; done
python tools/grad_time.py 3.25 30.25
