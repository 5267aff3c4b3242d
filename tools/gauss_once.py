import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d
n = 32768
s = float(sys.argv[1]) if len(sys.argv) > 1 else 30.25
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
o = d.DeviceArray(n, n)
for _ in range(2):
    blk.gaussian(s, s, o)
d.sync()
