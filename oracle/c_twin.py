"""ctypes access to oracle/libtopo_oracle.so (C/OpenMP twin).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libtopo_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            from . import build_oracle
            build_oracle.build()
        _lib = C.CDLL(LIB)
        _lib.oracle_tpi_std.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib.oracle_sx.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_int, C.c_double, C.c_void_p]
    return _lib


def threads():
    return lib().oracle_threads()


def tpi_std(dem, size, want_tpi=True, want_std=True, out_tpi=None, out_std=None):
    """out_tpi / out_std: float64 arrays to write into (a timing loop passes the same, already touched, arrays every
    time: a fresh 2 GB result array is faulted in page by page under the kernel's address-space lock, which on 256
    threads costs more than the computation)."""
    dem = np.ascontiguousarray(dem, dtype=np.float32)
    t = (out_tpi if out_tpi is not None else np.empty(dem.shape, np.float64)) if want_tpi else None
    s = (out_std if out_std is not None else np.empty(dem.shape, np.float64)) if want_std else None
    for a in (t, s):
        assert a is None or (a.shape == dem.shape and a.dtype == np.float64 and a.flags.c_contiguous)
    rc = lib().oracle_tpi_std(dem.ctypes.data, dem.shape[0], dem.shape[1], int(size),
                              t.ctypes.data if want_tpi else None, s.ctypes.data if want_std else None)
    assert rc == 0
    return t, s


def sx(dem, dj, di, dist, window, height):
    dem = np.ascontiguousarray(dem, dtype=np.float32)
    dj = np.ascontiguousarray(dj, dtype=np.int32)
    di = np.ascontiguousarray(di, dtype=np.int32)
    dist = np.ascontiguousarray(dist, dtype=np.float64)
    out = np.empty_like(dem)
    rc = lib().oracle_sx(dem.ctypes.data, dem.shape[0], dem.shape[1], dj.ctypes.data, di.ctypes.data,
                         dist.ctypes.data, len(dj), int(window), float(height), out.ctypes.data)
    assert rc == 0
    return out
