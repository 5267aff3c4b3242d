python -m pytest tests/test_gpu_parity.py tests/test_gpu_blocks.py -m gpu -q -x -k "gauss or gradient or aspect or nan or config3 or config5 or blocks" 2>&1 | tail -5
TOPO_AMD_GAUSS_F16=1 python tools/gauss_f16_error.py 2.25 8.0 30.25 2>&1 | cut -c1-200
for mt in 2 1; do TOPO_AMD_GAUSS_F16_MT=$mt python tools/gauss_axes_time.py 1.25 3.25 8.0 15.0 30.25 | sed "s/^/mt=$mt: /"; done
python tools/grad_time.py 1.25 3.25 8.0 15.0 30.25
