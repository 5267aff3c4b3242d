"""Time STD and TPI + STD on the 32768^2 bench DEM (median of 6 launches, HIP events) and CRC the outputs at
8192^2.  The same-box A/B of two builds of the library: run once as is and once with TOPO_AMD_LIBRARY=<the other .so>
(the switch TOPO_AMD_STD_RING_MIN this tool once flipped is a constant since round 5)."""
import json
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [31, 45, 65, 67]
out = {"library": os.environ.get("TOPO_AMD_LIBRARY", "in-tree")}
n = 8192
for integer in (True, False):
    dem = d.synth_dem(n, n, seed=0, integer=integer)
    blk = d.Block(dem)
    t, s = d.DeviceArray(n, n), d.DeviceArray(n, n)
    for size in sizes:
        blk.tpi_std(size, tpi=t, std=s)
        d.sync()
        tag = "int" if integer else "frac"
        out[f"crc_{tag}_{size}"] = [zlib.crc32(t.to_host().tobytes()), zlib.crc32(s.to_host().tobytes())]
        blk.tpi_std(size, std=s)
        d.sync()
        out[f"crc_{tag}_{size}"].append(zlib.crc32(s.to_host().tobytes()))
    for a in (t, s, dem):
        a.free()
n = 32768
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
t, s = d.DeviceArray(n, n), d.DeviceArray(n, n)
for size in sizes:
    ms = sorted(d.time_launches(lambda: blk.tpi_std(size, std=s), 6))
    out[f"std_{size}"] = round(ms[len(ms) // 2], 3)
    ms = sorted(d.time_launches(lambda: blk.tpi_std(size, tpi=t, std=s), 6))
    out[f"tpi_std_{size}"] = round(ms[len(ms) // 2], 3)
print(json.dumps(out))
