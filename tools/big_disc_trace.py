"""Launch topo tpi / std at 151, 401 and 1001 px on an 8192^2 DEM a few times: the workload for
`rocprofv3 --kernel-trace` when looking at the prefix-plane kernels (disc_big.hip)."""
import os, sys
sys.path.insert(0, os.getcwd())
from topo_descriptors_amd import device as d
n = 8192
dem = d.synth_dem(n, n, seed=0)
o = [d.DeviceArray(n, n) for _ in range(2)]
blk = d.Block(dem)
for size in (151, 401, 1001):
    for _ in range(2):
        blk.tpi_std(size, tpi=o[0])
        blk.tpi_std(size, std=o[1])
    d.sync()
