import json, os, sys
sys.path.insert(0, os.getcwd())
from topo_descriptors_amd import device as d
n = 32768
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
o = d.DeviceArray(n, n)
med = lambda f: round(sorted(d.time_launches(f, 7))[3], 3)
print(os.environ.get("TOPO_AMD_S1_DEBUG"), os.environ.get("TOPO_AMD_GAUSS_SPLIT_ONCE"), [med(lambda: blk.gaussian(0.0, s, o)) for s in (30.25, 16.0)])
