"""Time topo tpi per disc size on the bench DEM with fractional elevations (HIP events): run with
TOPO_AMD_TPI_FRACTION_MIN=1 (two marching passes for every size) and =999 (general kernel)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = 32768
out = {"TOPO_AMD_TPI_FRACTION_MIN": os.environ.get("TOPO_AMD_TPI_FRACTION_MIN")}
for integer in (False, True):
    dem = d.synth_dem(n, n, seed=0, integer=integer)
    blk = d.Block(dem)
    t = d.DeviceArray(n, n)
    row = {}
    for size in (5, 7, 11, 17, 25, 31, 45, 67, 101):
        blk.tpi_std(size, tpi=t)
        d.sync()
        d.timer_start()
        for _ in range(3):
            blk.tpi_std(size, tpi=t)
        row[size] = round(d.timer_stop() / 3, 3)
    out["integer_dem" if integer else "fractional_dem"] = row
    t.free()
    dem.free()
print(json.dumps(out))
