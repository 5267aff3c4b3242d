"""End-to-end time of the host-buffer call topo.tpi(numpy array): with the previous result still
alive (the result array gets fresh pages every time) and with it released first (the allocator
hands the same pages back).  TOPO_AMD_TRACE_HOST=1 adds the library's own phase timings;
TOPO_AMD_HOST_PREFAULT=0 switches the parallel pre-faulting of the result array off.
usage: host_trace.py [n=16384]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import topo  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(0)
dem = np.rint(1900 + 300 * rng.standard_normal((n, n))).astype(np.float32)
topo.tpi(dem[:512, :512], 67)
keep = []
for rep in range(4):
    t0 = time.perf_counter()
    keep.append(topo.tpi(dem, 67))
    dt = time.perf_counter() - t0
    print(f"{n}x{n} fresh result array, call {rep}: {dt * 1e3:7.1f} ms  {n * n / dt / 1e6:8.0f} Mpixels/s", flush=True)
del keep
for rep in range(4):
    t0 = time.perf_counter()
    out = topo.tpi(dem, 67)
    dt = time.perf_counter() - t0
    print(f"{n}x{n} recycled result pages, call {rep}: {dt * 1e3:7.1f} ms  {n * n / dt / 1e6:8.0f} Mpixels/s", flush=True)
    del out
