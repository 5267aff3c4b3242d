"""The fused Gaussian (sigma 3.25) and the gradient at sigma 3.25 on the bench DEM, one line per process: is there a slow mode?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from topo_descriptors_amd import device as d
n = 32768
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
o = [d.DeviceArray(n, n) for _ in range(4)]
med = lambda f: round(sorted(d.time_launches(f, 7))[3], 3)
print(os.environ.get("TOPO_AMD_GAUSS_TURN"), "gaussian 3.25", med(lambda: blk.gaussian(3.25, 3.25, o[0])), "gradient 3.25",
      med(lambda: blk.gradient(3.25, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3])))
