"""TPI + STD on the 32768^2 bench DEM with the STD plane shifted by a few offsets inside a larger allocation: do the two output
planes, 4 GiB apart when allocated back to back, collide in the memory channels?  (ms per launch, median of 6, HIP events)
usage: plane_offset_time.py [size=7]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402


class View:
    def __init__(self, ptr):
        self.ptr = ptr


size = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n = 32768
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
t = d.DeviceArray(n, n)
big = d.DeviceArray(n + 64, n)  # 8 MiB of slack
out = {"size": size}
for off in (0, 256, 1024, 4096, 65536, 131072 + 256, 1048576 + 4096, 2097152 + 8192 + 256):
    s = View(big.ptr + off)
    ms = sorted(d.time_launches(lambda: blk.tpi_std(size, tpi=t, std=s), 6))
    out[f"std_plane_plus_{off}"] = round(ms[len(ms) // 2], 3)
ms = sorted(d.time_launches(lambda: blk.tpi_std(size, std=View(big.ptr)), 6))
out["std_alone"] = round(ms[len(ms) // 2], 3)
print(json.dumps(out))
