"""Valley index of the reference's example script scale by scale (8192^2 at 100 m): kernel side, route, seconds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from topo_descriptors_amd import topo, helpers as hlp, device as d
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = d.synth_dem(n, n, seed=0)
dem = dev.to_host()
dev.free()
for scale in (1000, 2000, 4000, 6000, 10000, 20000, 30000, 60000, 100000):
    px = int(round(scale / 100.0))
    size = px + 1 - px % 2 if px % 2 == 0 else px
    t0 = time.time()
    out = topo.valley_ridge(dem, size, mode="valley", flat_list=[0, 0.2, 0.4])
    dt = time.time() - t0
    print(f"scale {scale:6d} m  kernel side {size:5d} px  {dt:7.2f} s", flush=True)
