"""Time topo tpi on the 32768^2 bench DEM for a few disc sizes (HIP events, 10 launches each)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [7, 17, 31, 67]
out = {"TOPO_AMD_TPI_RING_MIN": os.environ.get("TOPO_AMD_TPI_RING_MIN")}
n = 32768
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
t = d.DeviceArray(n, n)
for size in sizes:
    blk.tpi_std(size, tpi=t)
    d.sync()
    reps = 10
    d.timer_start()
    for _ in range(reps):
        blk.tpi_std(size, tpi=t)
    out[f"ms_{size}"] = round(d.timer_stop() / reps, 3)
print(json.dumps(out))
