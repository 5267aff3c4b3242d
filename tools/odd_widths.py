"""Every descriptor on DEM widths that are and are not multiples of 4 (16384 rows): the unaligned
widths go through the re-pitched copy of disc.hip.  usage: odd_widths.py"""
import os, sys
sys.path.insert(0, os.getcwd())
from topo_descriptors_amd import device as d
ny = 16384
for nx in (16384, 16383, 16382):
    a = d.synth_dem(ny, nx, seed=0)
    o = [d.DeviceArray(ny, nx) for _ in range(4)]
    blk = d.Block(a)
    def t(fn):
        fn(); d.sync(); d.timer_start(); fn(); return d.timer_stop()
    r = [f"nx {nx}:"]
    for size in (7, 67):
        r.append(f"tpi{size} {t(lambda: blk.tpi_std(size, tpi=o[0])):.2f}")
        r.append(f"std{size} {t(lambda: blk.tpi_std(size, std=o[1])):.2f}")
    for sigma in (3.25, 30.25):
        r.append(f"grad{sigma} {t(lambda: blk.gradient(sigma, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3])):.2f}")
    w, dj, di, dist = d.sx_offsets(0.0, 500.0, 30.0, -30.0)
    r.append(f"sx500 {t(lambda: blk.sx(dj, di, dist, w, 10.0, o[0])):.2f}")
    print(" ".join(r), "ms", flush=True)
    for x in o: x.free()
    a.free()
