"""The descriptors of BASELINE configs 2 / 3 at their OWN sizes, three calls each, for a rocprofv3 kernel trace
(tools/trace_window.py prints the launches): 8192^2 TPI / STD / TPI + STD at 7 and 65 px, 16384^2 gradient sigma 30.25.
usage: rocprofv3 --kernel-trace -d <dir> -- python3 tools/config_size_trace.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = 8192
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
t, s = d.DeviceArray(n, n), d.DeviceArray(n, n)
for size in (7, 65):
    for _ in range(3):
        blk.tpi_std(size, tpi=t)
    d.sync()
    for _ in range(3):
        blk.tpi_std(size, std=s)
    d.sync()
    for _ in range(3):
        blk.tpi_std(size, tpi=t, std=s)
    d.sync()
t.free(), s.free(), dem.free()
