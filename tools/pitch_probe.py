"""The disc kernels on DEMs of 32768 rows whose row pitch is and is not a power of two: ms per launch and per gigapixel
(median of 5, HIP events).  usage: pitch_probe.py [sizes...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [7, 17, 67]
ny = 32768
out = {}
for nx in (32768, 32768 + 64, 32768 + 256, 32768 - 1024):
    dem = d.synth_dem(ny, nx, seed=0)
    blk = d.Block(dem)
    t, s = d.DeviceArray(ny, nx), d.DeviceArray(ny, nx)
    gpx = ny * nx / 1e9
    for size in sizes:
        row = {}
        for name, fn in (("tpi", lambda: blk.tpi_std(size, tpi=t)), ("std", lambda: blk.tpi_std(size, std=s)),
                         ("tpi_std", lambda: blk.tpi_std(size, tpi=t, std=s))):
            ms = sorted(d.time_launches(fn, 5))[2]
            row[name] = [round(ms, 3), round(ms / gpx, 3)]
        out[f"nx{nx}_s{size}"] = row
    for a in (t, s, dem):
        a.free()
print(json.dumps(out, indent=0))
