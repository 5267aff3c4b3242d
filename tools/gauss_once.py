import os, sys
sys.path.insert(0, os.getcwd())
from topo_descriptors_amd import device as d
n = 32768
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
o = d.DeviceArray(n, n)
for _ in range(2):
    blk.gaussian(30.25, 30.25, o)
d.sync()
