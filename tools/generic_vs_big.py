"""Sizes outside the wave-shift kernels (even, 1, 2, > 101): LDS-gather kernel against the
prefix-plane path (TOPO_AMD_DISC_BIG_MIN_SIZE=1 forces the latter).  usage: generic_vs_big.py [n=8192]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
o = [d.DeviceArray(n, n) for _ in range(2)]
print("TOPO_AMD_DISC_BIG_MIN_SIZE =", os.environ.get("TOPO_AMD_DISC_BIG_MIN_SIZE", "(default)"))
for integer in (True, False):
    dem = d.synth_dem(n, n, seed=0, integer=integer)
    blk = d.Block(dem)
    for size in (2, 4, 8, 16, 32, 48, 66, 84, 100, 103, 111, 121):
        res = []
        for kw in (dict(tpi=o[0]), dict(std=o[1]), dict(tpi=o[0], std=o[1])):
            blk.tpi_std(size, **kw)
            d.sync()
            d.timer_start()
            blk.tpi_std(size, **kw)
            res.append(d.timer_stop())
        print(f"{'int ' if integer else 'frac'} size {size:4d}: tpi {res[0]:7.2f} std {res[1]:7.2f} both {res[2]:7.2f}", flush=True)
    dem.free()
