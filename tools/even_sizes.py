"""TPI / STD on disc sizes the wave-shift kernels do not cover (1, even sizes, beyond 101) on a 16384^2
DEM.  usage: even_sizes.py"""
import os, sys
sys.path.insert(0, os.getcwd())
from topo_descriptors_amd import device as d
n = 16384
a = d.synth_dem(n, n, seed=0)
b = d.DeviceArray(n, n)
c = d.DeviceArray(n, n)
blk = d.Block(a)
for size in (1, 2, 4, 6, 8, 16, 32, 66, 103, 151):
    blk.tpi_std(size, tpi=b); d.sync()
    d.timer_start(); blk.tpi_std(size, tpi=b); t1 = d.timer_stop()
    blk.tpi_std(size, std=c); d.sync()
    d.timer_start(); blk.tpi_std(size, std=c); t2 = d.timer_stop()
    print(f"{n}^2 size {size}: tpi {t1:.2f} ms, std {t2:.2f} ms", flush=True)
