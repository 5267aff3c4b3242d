"""Time topo std and tpi+std per disc size on the bench DEM (HIP events), for choosing
TOPO_AMD_STD_MARCH_MIN: run once with TOPO_AMD_STD_MARCH_MIN=1 (marching kernels for every
size) and once with =999 (general kernel for every size)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = 32768
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
t, s = d.DeviceArray(n, n), d.DeviceArray(n, n)
out = {"TOPO_AMD_STD_MARCH_MIN": os.environ.get("TOPO_AMD_STD_MARCH_MIN")}
for size in (5, 7, 11, 17, 25, 31, 45, 65, 67, 101):
    row = {}
    for name, fn in (("std", lambda: blk.tpi_std(size, std=s)), ("tpi_std", lambda: blk.tpi_std(size, tpi=t, std=s))):
        fn()
        d.sync()
        d.timer_start()
        for _ in range(3):
            fn()
        row[name] = round(d.timer_stop() / 3, 3)
    out[size] = row
print(json.dumps(out))
