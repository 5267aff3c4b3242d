"""Time topo tpi per disc size on the bench DEM (HIP events): run with TOPO_AMD_TPI_MARCH_MIN=1
(marching kernel for every size) and =999 (general kernel for every size)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = 32768
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
t = d.DeviceArray(n, n)
out = {"TOPO_AMD_TPI_MARCH_MIN": os.environ.get("TOPO_AMD_TPI_MARCH_MIN")}
for size in (5, 7, 17, 31, 45, 67, 77, 79, 85, 93, 101):
    blk.tpi_std(size, tpi=t)
    d.sync()
    d.timer_start()
    for _ in range(3):
        blk.tpi_std(size, tpi=t)
    out[size] = round(d.timer_stop() / 3, 3)
print(json.dumps(out))
