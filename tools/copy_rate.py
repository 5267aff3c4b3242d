"""What a plain device-to-device copy of the bench DEM reaches (the practical ceiling of a 1:1
read / write stream), next to TPI on the smallest discs.  usage: copy_rate.py"""
import os, sys
sys.path.insert(0, os.getcwd())
from topo_descriptors_amd import _lib, device as d
n = 32768
a = d.synth_dem(n, n, seed=0)
b = d.DeviceArray(n, n)
lib = _lib.lib()
for rep in range(3):
    d.timer_start()
    for _ in range(5):
        _lib.check(lib.topo_amd_memcpy_d2d(b.ptr, a.ptr, n * n * 4), "d2d")
    ms = d.timer_stop() / 5
    print(f"hipMemcpyAsync D2D of {n*n*4/1e9:.2f} GB: {ms:.3f} ms  -> {2*n*n*4/ms/1e6:.0f} GB/s read+write")
blk = d.Block(a)
for size in (3, 5, 7, 9, 13, 17):
    blk.tpi_std(size, tpi=b); d.sync()
    d.timer_start()
    for _ in range(5):
        blk.tpi_std(size, tpi=b)
    ms = d.timer_stop() / 5
    print(f"tpi size {size}: {ms:.3f} ms -> {2*n*n*4/ms/1e6:.0f} GB/s")
