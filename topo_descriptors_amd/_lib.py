"""ctypes binding of libtopo_amd.so (C ABI: include/topo_amd.h).  No fallback of any kind."""
import ctypes as C
import os
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# TOPO_AMD_LIBRARY: another build of the same library (A/B runs of two builds inside one GPU session)
LIB_PATH = os.environ.get("TOPO_AMD_LIBRARY") or os.path.join(HERE, "libtopo_amd.so")

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/topo_amd.h declares
SIGNATURES = {
    "topo_amd_version": (C.c_char_p, []),
    "topo_amd_last_error": (C.c_char_p, []),
    "topo_amd_device_count": (C.c_int, []),
    "topo_amd_init": (C.c_int, [C.c_int]),
    "topo_amd_shutdown": (C.c_int, []),
    "topo_amd_device_name": (C.c_int, [C.c_char_p, C.c_int]),
    "topo_amd_malloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "topo_amd_free": (C.c_int, [_vp]),
    "topo_amd_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "topo_amd_host_free": (C.c_int, [C.c_void_p]),
    "topo_amd_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_size_t]),
    "topo_amd_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_size_t]),
    "topo_amd_memcpy_d2d": (C.c_int, [_vp, _vp, C.c_size_t]),
    "topo_amd_memset": (C.c_int, [_vp, C.c_int, C.c_size_t]),
    "topo_amd_sync": (C.c_int, []),
    "topo_amd_release_host_planes": (C.c_int, []),
    "topo_amd_host_chunks": (C.c_int, [_i32p]),
    "topo_amd_valley_route": (C.c_int, [_i32p]),
    "topo_amd_dem_changed": (C.c_int, [_vp, C.c_size_t]),
    "topo_amd_raster_scan_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(C.c_uint64), _f32p]),
    "topo_amd_raster_class_from_scan": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64), _f32p]),
    "topo_amd_raster_class_set": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]),
    "topo_amd_raster_class_get": (C.c_int, [_vp, C.c_int, C.c_int, _i32p, _i32p, _f32p, _f32p, _f32p]),
    "topo_amd_shard_classify": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "topo_amd_timer_start": (C.c_int, []),
    "topo_amd_timer_stop": (C.c_int, [_f32p]),
    "topo_amd_mark": (C.c_int, [C.c_int]),
    "topo_amd_mark_elapsed": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "topo_amd_cu_count": (C.c_int, []),
    "topo_amd_synth_dem_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int]),
    "topo_amd_disc_tap_count": (C.c_int, [C.c_int]),
    "topo_amd_disc_mask": (C.c_int, [C.c_int, _f32p]),
    "topo_amd_halo_rows": (C.c_int, [C.c_int, C.c_double, C.c_double, _i32p, _i32p]),
    "topo_amd_tpi_std_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, _vp, _vp]),
    "topo_amd_gaussian_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                        C.c_double, C.c_int, C.c_int, _vp]),
    "topo_amd_sobel_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     _vp, _vp]),
    "topo_amd_gradient_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                        C.c_double, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp,
                                        _vp, _vp]),
    "topo_amd_sx_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p,
                                  C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _vp]),
    "topo_amd_tpi_multi_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, C.c_int, C.c_int,
                                         C.POINTER(_vp)]),
    "topo_amd_sx_multi_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p,
                                        _f64p, _i32p, C.c_double, C.c_int, C.c_int, C.POINTER(_vp)]),
    "topo_amd_valley_ridge_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _i32p, _vp, C.c_int,
                                            C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, _vp, _vp]),
    "topo_amd_mean_std_dev": (C.c_int, [_vp, C.c_size_t, _f64p, _f64p]),
    "topo_amd_valley_ridge_f32": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _i32p, _vp, C.c_int, C.c_int,
                                            C.c_double, C.c_double, _vp, _vp]),
    "topo_amd_tpi_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_double, _vp]),
    "topo_amd_std_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_double, _vp]),
    "topo_amd_tpi_std_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_double, _vp, _vp]),
    "topo_amd_tpi_std_multi_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _i32p, _f64p, _vp, _vp]),
    "topo_amd_gauss_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_double, _vp]),
    "topo_amd_sobel_f32": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "topo_amd_gradient_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                        _vp, _vp, _vp, _vp, _vp, _vp]),
    "topo_amd_sx_f32": (C.c_int, [_vp, C.c_int, C.c_int, _i32p, _i32p, _f64p, C.c_int, C.c_int,
                                  C.c_double, _vp]),
    "topo_amd_sx_multi_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p, _f64p, _i32p,
                                        C.c_double, C.POINTER(_vp)]),
    "topo_amd_shard_sx_multi": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p,
                                          _i32p, _f64p, _i32p, C.c_double, C.POINTER(_vp)]),
    "topo_amd_shard_valley_ridge": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _i32p, _vp, C.c_int,
                                              C.c_int, _vp, _vp]),
    "topo_amd_comm_unique_id": (C.c_int, [C.c_char_p]),
    "topo_amd_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_char_p]),
    "topo_amd_comm_rank": (C.c_int, []),
    "topo_amd_comm_size": (C.c_int, []),
    "topo_amd_comm_destroy": (C.c_int, []),
    "topo_amd_halo_exchange_start": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "topo_amd_halo_wait": (C.c_int, []),
    "topo_amd_gate_giveups": (C.c_int, [C.POINTER(C.c_uint)]),
    "topo_amd_shard_layout": (C.c_int, [C.c_int, C.c_int]),
    "topo_amd_shard_layout_get": (C.c_int, [_i32p, _i32p]),
    "topo_amd_shard_tpi_std": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "topo_amd_shard_gradient": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                          C.c_double, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "topo_amd_shard_sx": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p,
                                    C.c_int, C.c_int, C.c_double, _vp]),
}

UNIQUE_ID_BYTES = 128
RES_SCALAR, RES_1D, RES_2D = 0, 1, 2
DESC_TPI, DESC_STD, DESC_GAUSS, DESC_GRADIENT, DESC_SOBEL, DESC_SX, DESC_VALLEY_RIDGE = range(7)


class TopoAmdError(RuntimeError):
    """An entry point of libtopo_amd.so returned a non-zero status."""


_lock = threading.Lock()
_lib = None
_ready = False


def load():
    """dlopen the library and attach signatures (no GPU needed for this step)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise TopoAmdError(
                    f"{LIB_PATH} is missing: build it with `python -m topo_descriptors_amd.build` "
                    "(needs hipcc). There is no CPU fallback.")
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
        return _lib


def check(status, what):
    if status != 0:
        msg = load().topo_amd_last_error().decode(errors="replace")
        raise TopoAmdError(f"{what} failed with status {status}: {msg}")


def lib():
    """The loaded library bound to a GPU; raises when there is none."""
    global _ready
    handle = load()
    if not _ready:
        device = int(os.environ.get("TOPO_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        if handle.topo_amd_device_count() < 1:
            raise TopoAmdError("no HIP device visible: topo_descriptors_amd needs an AMD GPU "
                               "(gfx950); there is no CPU fallback")
        check(handle.topo_amd_init(device), "topo_amd_init")
        _ready = True
    return handle


def ptr(array):
    """void* of a C-contiguous numpy array (or None)."""
    if array is None:
        return None
    return array.ctypes.data_as(_vp)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)
