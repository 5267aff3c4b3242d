import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from topo_descriptors_amd import device as d
n = 8192
host = d.synth_dem(n, n, seed=0).to_host()
host[:, : n // 3] = np.nan
dem = d.DeviceArray.from_host(host)
clean = d.synth_dem(n, n, seed=0)
o = [d.DeviceArray(n, n) for _ in range(2)]
med = lambda f: round(sorted(d.time_launches(f, 5))[2], 3)
for size in (7, 67):
    row = {"size": size}
    for name, src in (("clean", clean), ("nan_third", dem)):
        blk = d.Block(src)
        row[f"tpi_{name}_ms"] = med(lambda: blk.tpi_std(size, tpi=o[0]))
        row[f"std_{name}_ms"] = med(lambda: blk.tpi_std(size, std=o[1]))
    print(json.dumps(row), flush=True)
window, dj, di, dist = d.sx_offsets(0.0, 500.0, 30.0, -30.0)
row = {}
for name, src in (("clean", clean), ("nan_third", dem)):
    blk = d.Block(src)
    row[f"sx_{name}_ms"] = med(lambda: blk.sx(dj, di, dist, window, 10.0, o[0]))
print(json.dumps(row))
