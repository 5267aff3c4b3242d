"""Host-buffer TPI at several scales: a loop of topo.tpi calls (one upload each) against one
topo.tpi_std_multi call (one upload for all).  usage: multi_scale_time.py [n=16384]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import topo  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(0)
dem = np.rint(1500.0 + 300.0 * rng.random((n, n), dtype=np.float32)).astype(np.float32)
sizes = [7, 17, 33, 67]
topo.tpi(dem[:512, :512], 7)  # initialise
for rep in range(2):
    t0 = time.perf_counter()
    loop = [topo.tpi(dem, s) for s in sizes]
    t1 = time.perf_counter()
    multi, _ = topo.tpi_std_multi(dem, sizes, want_std=False)
    t2 = time.perf_counter()
    same = all(np.array_equal(a, b) for a, b in zip(loop, multi))
    print(f"{n}^2, TPI at {sizes}: loop {1e3 * (t1 - t0):.0f} ms, one call {1e3 * (t2 - t1):.0f} ms, identical planes: {same}", flush=True)
    del loop, multi
