"""Launch topo tpi at 67 px on the bench DEM (whole metres, then fractional elevations) a few times:
the workload for `rocprofv3 --kernel-trace` / tools/pmc_passes.sh when looking at the TPI kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
size = int(sys.argv[2]) if len(sys.argv) > 2 else 67
t = d.DeviceArray(n, n)
for integer in (True, False):
    dem = d.synth_dem(n, n, seed=0, integer=integer)
    blk = d.Block(dem)
    for _ in range(6):
        blk.tpi_std(size, tpi=t)
    d.sync()
    dem.free()
