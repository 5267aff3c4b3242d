#!/usr/bin/env python3
"""One host-buffer TPI 67 px call on page-locked arrays (after a warm-up call), for a trace:
    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/pipe_trace -- python3 tools/host_pipeline_once.py [n=16384]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
if "torch" in sys.argv:  # (bench.py imports torch first: the library then binds to the HIP runtime torch ships)
    import torch  # noqa: F401
lib = _lib.lib()
if "warm_pageable" in sys.argv:
    from topo_descriptors_amd import topo
    topo.tpi(np.zeros((4096, 4096), np.float32), 67)
if "bench" in sys.argv:  # exactly what bench.py's end_to_end section does
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from topo_descriptors_amd import device as d
    print(bench.end_to_end(d, _lib, 67))
    sys.exit(0)
if "big_pageable" in sys.argv:
    from topo_descriptors_amd import topo
    w = topo.tpi(np.rint(1900.0 + 300.0 * np.random.default_rng(1).standard_normal((n, n))).astype(np.float32), 67)
    del w
dem = np.rint(1900.0 + 300.0 * np.random.default_rng(0).standard_normal((n, n))).astype(np.float32)
hin, hout = C.c_void_p(), C.c_void_p()
_lib.check(lib.topo_amd_host_alloc(C.byref(hin), dem.nbytes), "host_alloc")
_lib.check(lib.topo_amd_host_alloc(C.byref(hout), dem.nbytes), "host_alloc")
np.frombuffer((C.c_char * dem.nbytes).from_address(hin.value), dtype=np.float32)[:] = dem.ravel()
for k in range(3):
    t0 = time.perf_counter()
    _lib.check(lib.topo_amd_tpi_f32(hin, n, n, 67, 0.0, hout), "tpi")
    print(f"call {k}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
