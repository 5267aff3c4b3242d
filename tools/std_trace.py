"""Launch STD (and TPI + STD) at one size on the bench DEM a few times: workload for rocprofv3 --kernel-trace --stats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
size = int(sys.argv[2]) if len(sys.argv) > 2 else 67
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
t, s = d.DeviceArray(n, n), d.DeviceArray(n, n)
for _ in range(4):
    blk.tpi_std(size, std=s)
for _ in range(4):
    blk.tpi_std(size, tpi=t, std=s)
d.sync()
