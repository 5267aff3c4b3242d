"""Launch topo sx (azimuth 0, radius 500 m and 2000 m on a 30 m grid) on the bench DEM a few
times: the workload for `rocprofv3 --kernel-trace` / tools/pmc_passes.sh when looking at the Sx
kernel.   usage: sx_trace.py [n=32768] [radii=500,2000]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
radii = [float(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "500,2000").split(",")]
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
out = d.DeviceArray(n, n)
for radius in radii:
    window, dj, di, dist = d.sx_offsets(0.0, radius, 30.0, -30.0)
    for _ in range(4):
        blk.sx(dj, di, dist, window, 10.0, out)
    d.sync()
