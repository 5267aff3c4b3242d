"""Look for performance cliffs: gradient over sigma and sig_ratio, Gaussian, Sx over radius, valley
index over size, on one DEM (HIP events, second call).  usage: param_sweep.py [n=8192]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d, topo  # noqa: E402

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
for sigma in (0.75, 1.25, 2.25, 3.25, 4.5, 5.5, 5.9, 6.1, 8.0, 12.0, 20.0, 30.25, 60.0, 100.0, 107.0, 109.0, 120.0):
    ms = t(lambda: blk.gradient(sigma, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3]))
    print(f"gradient sigma {sigma:6.2f}: {ms:8.2f}", flush=True)
for sigma, ratio in ((3.25, 2.0), (3.25, 0.5), (30.25, 2.0), (30.25, 0.25)):
    ms = t(lambda: blk.gradient(sigma, [30.0], [-30.0], sig_ratio=ratio, dx=o[0], dy=o[1], slope=o[2], aspect=o[3]))
    print(f"gradient sigma {sigma:6.2f} sig_ratio {ratio}: {ms:8.2f}", flush=True)
for sigma in (0.75, 3.25, 30.25, 120.0):
    print(f"gaussian sigma {sigma:6.2f}: {t(lambda: blk.gaussian(sigma, sigma, o[0])):8.2f}", flush=True)
for radius in (100.0, 500.0, 1000.0, 2000.0, 3000.0, 4000.0, 6000.0):
    w, dj, di, dist = d.sx_offsets(0.0, radius, 30.0, -30.0)
    print(f"sx radius {radius:6.0f} (window {w}): {t(lambda: blk.sx(dj, di, dist, w, 10.0, o[0])):8.2f}", flush=True)
mean, stdev = d.mean_std(dem)
for size in (3, 7, 17, 33, 67, 101):
    try:
        taps, ksize, angles = topo._valley_ridge_tables(topo._valley_kernels(size, [0, 0.15, 0.3]),
                                                        np.arange(0, 180, dtype=np.float32))
        ms = t(lambda: blk.valley_ridge(taps, ksize, angles, 3, mean, stdev, o[0], o[1]))
        print(f"valley index size {size:3d} (largest rotated kernel {ksize.max()}): {ms:10.2f}", flush=True)
    except Exception as exc:  # noqa: BLE001
        print(f"valley index size {size:3d}: {type(exc).__name__}: {str(exc)[:150]}", flush=True)
