#!/usr/bin/env python3
"""Host-buffer calls (upload || kernels || download in row chunks, csrc/capi.hip run_pipelined) against the serial order:
topo_amd_tpi_f32 / topo_amd_gradient_f32 on an n x n DEM, page-locked and pageable arrays, for a few chunk sizes.
Each configuration runs in a child process (a fresh set of device planes; the switches themselves are read at every call).

    python tools/host_pipeline_time.py [n=16384]
"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child(n):
    from topo_descriptors_amd import _lib
    lib = _lib.lib()
    dem = np.rint(1900.0 + 300.0 * np.random.default_rng(0).standard_normal((n, n))).astype(np.float32)

    def pinned(count):
        out = []
        for _ in range(count):
            h = C.c_void_p()
            _lib.check(lib.topo_amd_host_alloc(C.byref(h), dem.nbytes), "host_alloc")
            out.append(h)
        return out

    hin, *houts = pinned(5)
    np.frombuffer((C.c_char * dem.nbytes).from_address(hin.value), dtype=np.float32)[:] = dem.ravel()
    pageable_out = [np.empty_like(dem) for _ in range(4)]
    for o in pageable_out:
        o[:] = 0
    rx, ry = np.array([30.0]), np.array([-30.0])

    def best(fn, reps=3):
        b = None
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            b = dt if b is None else min(b, dt)
        return b * 1e3

    res = {}
    res["tpi67 pinned"] = best(lambda: _lib.check(lib.topo_amd_tpi_f32(hin, n, n, 67, 0.0, houts[0]), "tpi"))
    res["tpi67 pageable"] = best(lambda: _lib.check(lib.topo_amd_tpi_f32(_lib.ptr(dem), n, n, 67, 0.0, _lib.ptr(pageable_out[0])), "tpi"))
    res["gradient 3.25 pinned (4 planes)"] = best(lambda: _lib.check(lib.topo_amd_gradient_f32(
        hin, n, n, 3.25, 1.0, 0, _lib.ptr(rx), _lib.ptr(ry), *houts), "gradient"))
    res["gradient 3.25 pageable (4 planes)"] = best(lambda: _lib.check(lib.topo_amd_gradient_f32(
        _lib.ptr(dem), n, n, 3.25, 1.0, 0, _lib.ptr(rx), _lib.ptr(ry), *[_lib.ptr(o) for o in pageable_out]), "gradient"))
    for k, v in res.items():
        print(f"    {k:36s} {v:8.2f} ms")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    if len(sys.argv) > 2 and sys.argv[2] == "child":
        child(n)
        return
    for env in ({"TOPO_AMD_HOST_PIPELINE": "0"}, {}, {"TOPO_AMD_HOST_DOWNLOADS": "thread"},
                {"TOPO_AMD_HOST_DOWNLOADS": "inline"}, {"TOPO_AMD_HOST_CHUNK_MB": "128"}, {"TOPO_AMD_HOST_CHUNK_MB": "32"}):
        print(env or "default (64 MB chunks)", flush=True)
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), str(n), "child"], env=e, check=False)


if __name__ == "__main__":
    main()
