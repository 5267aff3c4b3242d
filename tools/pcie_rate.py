#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer API (upload + kernel + download), for DESIGN.md.
Never the bench `value`.  Usage on the GPU box: python tools/pcie_rate.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import topo  # noqa: E402

rng = np.random.default_rng(0)
for n in (4096, 8192, 16384, 32768):
    dem = np.rint(1900 + 300 * rng.standard_normal((n, n))).astype(np.float32)
    for size in (7, 67):
        warm = topo.tpi(dem, size)
        t0 = time.perf_counter()
        out = topo.tpi(dem, size)   # held, as a caller would: releasing 1 GiB costs as much as a copy
        dt = time.perf_counter() - t0
        del warm, out
        print(f"topo.tpi host-buffer {n}x{n} size {size}: {dt*1e3:8.1f} ms  {n*n/dt/1e6:9.0f} Mpixels/s "
              f"({2*dem.nbytes/dt/1e9:5.1f} GB/s over PCIe both copies, pageable arrays)")
