"""Valley index by FFT against the direct kernel (same tables), and timings of both.
usage: valley_fft_check.py [n=4096]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d, topo  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
mean, stdev = d.mean_std(dem)
o = [d.DeviceArray(n, n) for _ in range(4)]
angles = np.arange(0, 180, dtype=np.float32)


def run(fft, taps, ksize, ang, norm, direction):
    os.environ["TOPO_AMD_VALLEY_FFT_MIN_KERNEL"] = "1" if fft else "100000"
    blk.valley_ridge(taps, ksize, ang, 3, mean, stdev, norm, direction)
    d.sync()
    d.timer_start()
    blk.valley_ridge(taps, ksize, ang, 3, mean, stdev, norm, direction)
    return d.timer_stop()


for size in (33, 45, 67, 101, 151, 301, 601):
    t0 = time.perf_counter()
    taps, ksize, ang = topo._valley_ridge_tables(topo._valley_kernels(size, [0, 0.15, 0.3]), angles)
    host = time.perf_counter() - t0
    ms_fft = run(True, taps, ksize, ang, o[0], o[1])
    line = f"size {size:4d} (kernels up to {ksize.max():4d} px, host tables {host:5.1f} s): fft {ms_fft:9.1f} ms"
    if ksize.max() <= 100 and n <= 4096 or size <= 33:
        ms_dir = run(False, taps, ksize, ang, o[2], o[3])
        a, b = o[0].to_host(), o[2].to_host()
        da, db = o[1].to_host(), o[3].to_host()
        line += (f", direct {ms_dir:9.1f} ms; norm max|diff| {np.max(np.abs(a - b)):.2e} of max {b.max():.3f}; "
                 f"direction differs on {100.0 * np.mean(da != db):.3f} % of pixels")
    print(line, flush=True)
