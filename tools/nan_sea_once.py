import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from topo_descriptors_amd import device as d
n = 8192
host = d.synth_dem(n, n, seed=0).to_host()
host[:, : n // 3] = np.nan
dem = d.DeviceArray.from_host(host)
o = d.DeviceArray(n, n)
blk = d.Block(dem)
s = float(sys.argv[1]) if len(sys.argv) > 1 else 30.25
for _ in range(3):
    blk.gaussian(s, s, o)
d.sync()
