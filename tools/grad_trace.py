"""Launch topo gradient (sigma 3.25 and 30.25, 4 outputs and slope+aspect only) on the bench DEM a
few times: the workload for `rocprofv3 --kernel-trace` / tools/pmc_passes.sh when looking at the
Gaussian / gradient kernels.   usage: grad_trace.py [n=32768] [sigmas=3.25,30.25]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
sigmas = [float(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "3.25,30.25").split(",")]
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
o = [d.DeviceArray(n, n) for _ in range(4)]
for sigma in sigmas:
    for _ in range(3):
        blk.gradient(sigma, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3])
    d.sync()
    for _ in range(3):
        blk.gradient(sigma, [30.0], [-30.0], slope=o[2], aspect=o[3])
    d.sync()
