"""TPI + STD at 7 px on the 32768^2 bench DEM with the two output planes taken from a pool of six allocations, pair by pair: does
the slow mode of the two-plane kernels (2.8 or 3.7 ms) belong to the planes?  (ms per launch, median of 5)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = 32768
size = int(sys.argv[1]) if len(sys.argv) > 1 else 7
pool = [d.DeviceArray(n, n) for _ in range(3)]
dem = d.synth_dem(n, n, seed=0)
pool += [d.DeviceArray(n, n) for _ in range(3)]
blk = d.Block(dem)
out = {"dem_ptr": hex(dem.ptr), "pool": [hex(p.ptr) for p in pool]}
for a in range(6):
    for b in range(6):
        if a == b:
            continue
        ms = sorted(d.time_launches(lambda: blk.tpi_std(size, tpi=pool[a], std=pool[b]), 5))
        out[f"tpi{a}_std{b}"] = round(ms[2], 3)
ms = sorted(d.time_launches(lambda: blk.tpi_std(size, std=pool[0]), 5))
out["std_alone_0"] = round(ms[2], 3)
print(json.dumps(out))
