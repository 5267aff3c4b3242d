"""Launch topo std / tpi+std (and tpi) on the bench DEM a few times: the workload for
`rocprofv3 --kernel-trace` / tools/pmc_passes.sh when looking at the STD kernels.
usage: std_trace.py [n=32768] [sizes=67,7]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "67,7").split(",")]
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
t, s = d.DeviceArray(n, n), d.DeviceArray(n, n)
for size in sizes:
    for _ in range(3):
        blk.tpi_std(size, std=s)
    d.sync()
    for _ in range(3):
        blk.tpi_std(size, tpi=t, std=s)
    d.sync()
    for _ in range(3):
        blk.tpi_std(size, tpi=t)
    d.sync()
