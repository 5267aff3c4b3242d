"""Gaussian on a raster whose ordinary values lie beyond the f16 kernels' +-1e5 (a DEM in millimetres), 8192^2: with the
route chosen by sampling (vector-ALU kernels) and with TOPO_AMD_GAUSS_LARGE_SAMPLE=0 (matrix-core kernels, every tile
repaired): ADVICE r03, medium (a)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d
n = 8192
dem = d.synth_dem(n, n, seed=0, scale=1000.0) if "scale" in d.synth_dem.__code__.co_varnames else None
if dem is None:
    import numpy as np
    from oracle import topo_oracle as orc
    dem = d.DeviceArray.from_host(orc.synthetic_dem(n, n, seed=0) * np.float32(1000.0))
blk = d.Block(dem)
o = d.DeviceArray(n, n)
med = lambda f: round(sorted(d.time_launches(f, 5))[2], 3)
print(os.environ.get("TOPO_AMD_GAUSS_LARGE_SAMPLE"), {s: med(lambda: blk.gaussian(s, s, o)) for s in (3.25, 13.0)})
