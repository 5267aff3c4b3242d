"""Launch the gradient at one sigma on the bench DEM a few times: the workload for rocprofv3 --kernel-trace --stats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 30.25
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
o = [d.DeviceArray(n, n) for _ in range(4)]
for _ in range(6):
    blk.gradient(sigma, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3])
d.sync()
