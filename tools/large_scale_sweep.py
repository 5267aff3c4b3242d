"""The scales of the reference's example script (100 m ... 100 km) on a 100 m grid: disc sizes up to
1001 px, Gaussian sigma up to 250.  ms per call on one resident DEM.  usage: large_scale_sweep.py [n=8192]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dem = d.synth_dem(n, n, seed=0)
o = [d.DeviceArray(n, n) for _ in range(4)]
blk = d.Block(dem)


def t(fn):
    fn()
    d.sync()
    d.timer_start()
    fn()
    return d.timer_stop()


print(f"{n}x{n}; ms per call")
for size in (101, 103, 151, 201, 301, 401, 601, 1001, 2001):
    a = t(lambda: blk.tpi_std(size, tpi=o[0]))
    b = t(lambda: blk.tpi_std(size, std=o[1]))
    print(f"disc {size:5d} px: tpi {a:9.1f}  std {b:9.1f}", flush=True)
for sigma in (25.0, 50.0, 75.0, 125.0, 250.0):
    g = t(lambda: blk.gradient(sigma, [100.0], [-100.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3]))
    s = t(lambda: blk.gaussian(sigma, sigma, o[0]))
    print(f"sigma {sigma:6.1f} (radius {int(4 * sigma + 0.5):4d}): gradient {g:9.1f}  gaussian {s:9.1f}", flush=True)
