for f in 1 0; do TOPO_AMD_GAUSS_F16_TALL=$f python tools/gauss_axes_time.py 30.25 | sed "s/^/tall=$f: /"; done
