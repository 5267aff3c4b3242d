"""Axis-0 Gaussian alone on the 32768^2 bench DEM (median of 7 launches): A/B of TOPO_AMD_GAUSS_SPLIT_ONCE, workload for counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from topo_descriptors_amd import device as d
n = 32768
sigmas = [float(a) for a in sys.argv[1:]] or [30.25, 16.0]
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
o = d.DeviceArray(n, n)
med = lambda f: round(sorted(d.time_launches(f, 7))[3], 3)
print(os.environ.get("TOPO_AMD_GAUSS_SPLIT_ONCE"), [med(lambda: blk.gaussian(s, 0.0, o)) for s in sigmas])
