python -m pytest tests/test_gpu_parity.py tests/test_gpu_blocks.py -m gpu -q -x -k "gauss or gradient or aspect or nan or config3 or config5 or blocks" 2>&1 | grep -E "passed|failed|Error|assert" | head
for m in 31 16; do TOPO_AMD_GAUSS_FUSED_MFMA_MAX_RADIUS=$m python tools/grad_time.py 3.25 5.0 7.0 | sed "s/^/fused max radius $m: /"; done
