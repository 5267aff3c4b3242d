"""Launch the valley index (7 px by default, 180 angles x 3 planes) on the bench DEM a few times: the workload for
rocprofv3 --kernel-trace --stats and for the PMC passes (PMC_SCRIPT=tools/valley_trace.py tools/pmc_passes.sh <dir> <n> <size>)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d, topo  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
size = int(sys.argv[2]) if len(sys.argv) > 2 else 7
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
o = [d.DeviceArray(n, n) for _ in range(2)]
taps, ksize, angles = topo._valley_ridge_tables(topo._valley_kernels(size, [0, 0.15, 0.3]), np.arange(0, 180, dtype=np.float32))
for _ in range(4):
    blk.valley_ridge(taps, ksize, angles, 3, 1500.0, 400.0, o[0], o[1])
d.sync()
print("route", d.valley_route())
