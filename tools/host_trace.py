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

# the device-resident path: DeviceArray.to_host() into a fresh numpy array
from topo_descriptors_amd import device as d  # noqa: E402

dev = d.DeviceArray.from_host(dem)
blk = d.Block(dev)
out_dev = d.DeviceArray(n, n)
keep = []
for rep in range(3):
    t0 = time.perf_counter()
    blk.tpi_std(67, tpi=out_dev)
    keep.append(out_dev.to_host())
    dt = time.perf_counter() - t0
    print(f"{n}x{n} resident DEM, kernel + to_host(), call {rep}: {dt * 1e3:7.1f} ms", flush=True)
