"""TPI / STD / both for every odd disc size 3 ... 101 on one DEM, whole metres and fractional:
look for steps between neighbouring sizes.  usage: size_sweep.py [n=16384]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
o = [d.DeviceArray(n, n) for _ in range(2)]


def t(fn):
    fn()
    d.sync()
    d.timer_start()
    fn()
    fn()
    return d.timer_stop() / 2


for integer in (True, False):
    dem = d.synth_dem(n, n, seed=0, integer=integer)
    blk = d.Block(dem)
    print(f"{n}x{n}, {'whole metres' if integer else 'fractional elevations'}; ms: size tpi std both")
    for size in range(3, 103, 2):
        a = t(lambda: blk.tpi_std(size, tpi=o[0]))
        b = t(lambda: blk.tpi_std(size, std=o[1]))
        c = t(lambda: blk.tpi_std(size, tpi=o[0], std=o[1]))
        print(f"{size:4d} {a:7.2f} {b:7.2f} {c:7.2f}", flush=True)
    dem.free()
