import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from topo_descriptors_amd import device as d
n = 32768
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
o = [d.DeviceArray(n, n) for _ in range(4)]
for _ in range(4):
    blk.gradient(3.25, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3])
d.sync()
