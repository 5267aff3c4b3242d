"""A/B of the TPI kernels on the bench DEM: timing at 32768^2 (HIP events) and a CRC of the output
at 8192^2, whole metres and fractional.  Run once per setting of TOPO_AMD_TPI_RING_MIN (5 = ring
build, 999 = tpi_march_kernel); equal CRCs = equal bits.  usage: ring_ab.py [sizes...]"""
import json
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [7, 17, 31, 67, 101]
out = {"TOPO_AMD_TPI_RING_MIN": os.environ.get("TOPO_AMD_TPI_RING_MIN")}
n = 8192
for integer in (True, False):
    dem = d.synth_dem(n, n, seed=0, integer=integer)
    blk = d.Block(dem)
    t = d.DeviceArray(n, n)
    for size in sizes:
        blk.tpi_std(size, tpi=t)
        d.sync()
        out[f"crc_{'int' if integer else 'frac'}_{size}"] = zlib.crc32(t.to_host().tobytes())
    t.free()
    dem.free()
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
