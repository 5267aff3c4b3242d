"""Device-resident DEM blocks: upload once, run many descriptors, download what you need.

Thin Python over the ``*_dev`` entry points of the C ABI (include/topo_amd.h).  This is what
``bench.py`` times (inputs already in HBM) and what the batch wrappers of the reference
(``compute_tpi`` ... topo.py:88-141) would use to keep the DEM on the GPU across scales.
"""
import ctypes as C

import numpy as np

from . import _lib


class DeviceArray:
    """rows x nx float32 plane in HBM owned by libtopo_amd."""

    def __init__(self, rows, nx):
        self.rows, self.nx = int(rows), int(nx)
        self.nbytes = self.rows * self.nx * 4
        p = C.c_void_p()
        _lib.check(_lib.lib().topo_amd_malloc(C.byref(p), self.nbytes), "topo_amd_malloc")
        self.ptr = p.value

    @classmethod
    def from_host(cls, array):
        a = _lib.as_f32(array)
        d = cls(a.shape[0], a.shape[1])
        _lib.check(_lib.lib().topo_amd_memcpy_h2d(d.ptr, _lib.ptr(a), d.nbytes), "memcpy_h2d")
        return d

    def row_ptr(self, row):
        return self.ptr + int(row) * self.nx * 4

    def to_host(self, row0=0, rows=None):
        rows = self.rows - row0 if rows is None else rows
        out = np.empty((rows, self.nx), dtype=np.float32)
        _lib.check(_lib.lib().topo_amd_memcpy_d2h(_lib.ptr(out), self.row_ptr(row0), out.nbytes),
                   "memcpy_d2h")
        return out

    def upload_rows(self, array, row0=0):
        a = _lib.as_f32(array)
        _lib.check(_lib.lib().topo_amd_memcpy_h2d(self.row_ptr(row0), _lib.ptr(a), a.nbytes),
                   "memcpy_h2d")

    def free(self):
        if self.ptr:
            _lib.check(_lib.lib().topo_amd_free(self.ptr), "topo_amd_free")
            self.ptr = None

    def __del__(self):  # best effort
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


def sync():
    _lib.check(_lib.lib().topo_amd_sync(), "topo_amd_sync")


class RasterScan:
    """The class of a raster that is held as several row blocks (include/topo_amd.h, "what kernel routing may know about
    a raster"): ``add`` the rows each block owns, then ``declare`` it for every block the descriptors will be called on -
    those blocks then take the kernels the whole raster takes.  A declaration belongs to the block's MEMORY (it goes when
    the library writes or frees it), not to a thread.  (A block that is the whole raster needs none of this, and
    ``ShardedDEM`` does it by itself.)"""

    def __init__(self):
        self.counts = (C.c_uint64 * 3)(0, 0, 0)
        self.range = (C.c_float * 2)(np.inf, -np.inf)

    def add(self, block, own_row0=None, own_rows=None):
        """``block``: a :class:`Block`; the rows it OWNS default to all of its rows (give them when blocks overlap)."""
        o0 = block.row0 if own_row0 is None else own_row0
        on = block.rows if own_rows is None else own_rows
        _lib.check(_lib.lib().topo_amd_raster_scan_dev(*block._head(), int(o0), int(on), self.counts, self.range),
                   "raster_scan_dev")
        return self

    def declare(self, *blocks):
        """Declare the class added up so far for the device rows of each :class:`Block` in ``blocks``."""
        if not blocks:
            raise ValueError("RasterScan.declare: name the blocks the class is declared for (it is keyed by their memory)")
        for blk in blocks:
            _lib.check(_lib.lib().topo_amd_raster_class_from_scan(blk.data.row_ptr(blk.first), blk.rows, blk.gny, blk.nx,
                                                                  self.counts, self.range), "raster_class_from_scan")
        return self


def forget_raster_class(block=None):
    """Withdraw what was declared for ``block``'s rows (``None``: every declaration): undeclared partial row blocks are
    ordinary DEMs in whole metres."""
    if block is None:
        _lib.check(_lib.load().topo_amd_raster_class_set(None, 0, 0, 0, -1, 0.0, 0.0, 0.0), "raster_class_set")
    else:
        _lib.check(_lib.load().topo_amd_raster_class_set(block.data.row_ptr(block.first), block.rows, block.gny, block.nx,
                                                         -1, 0.0, 0.0, 0.0), "raster_class_set")


def raster_class(block):
    """``(declared, large, lo, hi, frac_share)`` a call on ``block`` would take (``declared`` False: the ordinary DEM's)."""
    dec, large = C.c_int32(), C.c_int32()
    lo, hi, share = C.c_float(), C.c_float(), C.c_float()
    _lib.check(_lib.lib().topo_amd_raster_class_get(block.data.row_ptr(block.first), block.gny, block.nx, C.byref(dec),
                                                    C.byref(large), C.byref(lo), C.byref(hi), C.byref(share)),
               "raster_class_get")
    return bool(dec.value), bool(large.value), lo.value, hi.value, share.value


def release_host_planes():
    """Give the device planes the host-buffer calls (``topo.tpi(ndarray)`` ...) keep between calls back to the GPU: a
    32768 x 32768 gradient leaves about 20 GiB behind (they are kept because copies out of freshly mapped device memory run
    at half the link's rate)."""
    _lib.check(_lib.lib().topo_amd_release_host_planes(), "release_host_planes")


def host_chunks():
    """Row chunks the calling thread's last host-buffer call ran in (1: the serial order)."""
    n = C.c_int32()
    _lib.check(_lib.lib().topo_amd_host_chunks(C.byref(n)), "host_chunks")
    return n.value


def valley_route():
    """The evaluation the calling thread's last valley / ridge call took: 0 tap by tap (``csrc/valley.hip``), 1 matrix pipe
    (``csrc/valley_mfma.hip``), 2 FFT; + 4 when the tap-by-tap kernel followed the matrix pipe over its flagged tiles; + 8 when the
    matrix pipe ran its folded form (point-symmetric tables: pairs of opposite window cells); + 16 when that form streamed its pixel
    operands (kernels of 19 px and more)."""
    n = C.c_int32()
    _lib.check(_lib.lib().topo_amd_valley_route(C.byref(n)), "valley_route")
    return n.value


def dem_changed(array):
    """Tell the library that ``array`` (a :class:`DeviceArray`) was written by something other than the library."""
    _lib.check(_lib.lib().topo_amd_dem_changed(array.ptr, array.nbytes), "dem_changed")


def timer_start():
    _lib.check(_lib.lib().topo_amd_timer_start(), "timer_start")


def timer_stop():
    ms = C.c_float()
    _lib.check(_lib.lib().topo_amd_timer_stop(C.byref(ms)), "timer_stop")
    return ms.value


def mark(index):
    """Record numbered event ``index`` (0 .. 511) on the compute stream."""
    _lib.check(_lib.lib().topo_amd_mark(int(index)), "mark")


def mark_elapsed(a, b):
    """Milliseconds between marks ``a`` and ``b`` (waits for ``b``)."""
    ms = C.c_float()
    _lib.check(_lib.lib().topo_amd_mark_elapsed(int(a), int(b), C.byref(ms)), "mark_elapsed")
    return ms.value


def time_launches(fn, reps, warm=1):
    """Per-launch HIP-event durations (ms) of ``reps`` back-to-back calls of ``fn`` (<= 511), after
    ``warm`` untimed ones: one event between consecutive launches, no host synchronise inside."""
    if not 1 <= reps <= 511:
        raise ValueError(f"time_launches: reps = {reps} outside 1 .. 511 (the library keeps 512 numbered events)")
    for _ in range(warm):
        fn()
    sync()
    mark(0)
    for k in range(reps):
        fn()
        mark(k + 1)
    return [mark_elapsed(k, k + 1) for k in range(reps)]


def synth_dem(rows, nx, row0=0, seed=0, out=None, out_row=0, integer=True):
    """Fill (part of) a DeviceArray with the deterministic synthetic terrain.

    integer=True rounds to whole metres, integer=False keeps fractional elevations."""
    d = out if out is not None else DeviceArray(rows, nx)
    _lib.check(_lib.lib().topo_amd_synth_dem_dev(d.row_ptr(out_row), rows, row0, nx, seed, int(bool(integer))),
               "synth_dem")
    return d


class Block:
    """A device plane seen as rows [row0, row0+rows) of a global gny x nx DEM."""

    def __init__(self, data, row0=0, gny=None, first_buffer_row=0, rows=None):
        self.data = data
        self.first = first_buffer_row
        self.rows = data.rows - first_buffer_row if rows is None else rows
        self.row0 = row0
        self.gny = self.rows if gny is None else gny
        self.nx = data.nx

    def _head(self):
        return (self.data.row_ptr(self.first), self.rows, self.row0, self.gny, self.nx)

    def _range(self, out_row0, out_rows):
        o0 = self.row0 if out_row0 is None else out_row0
        on = (self.row0 + self.rows - o0) if out_rows is None else out_rows
        return o0, on

    def tpi_std(self, size, tpi=None, std=None, out_row0=None, out_rows=None):
        o0, on = self._range(out_row0, out_rows)
        _lib.check(_lib.lib().topo_amd_tpi_std_dev(*self._head(), int(size), o0, on,
                                                   tpi.ptr if tpi else None,
                                                   std.ptr if std else None), "tpi_std_dev")

    def tpi_multi(self, sizes, outs, out_row0=None, out_rows=None):
        """TPI for several disc sizes; pairs of small sizes (5 ... 11 px) share one pass over the DEM
        (``topo_amd_tpi_multi_dev``).  outs: one DeviceArray per size.  Same bits as ``tpi_std`` per size."""
        o0, on = self._range(out_row0, out_rows)
        sz = np.ascontiguousarray(np.atleast_1d(sizes), dtype=np.int32)
        if sz.size != len(outs):
            raise ValueError(f"{sz.size} sizes but {len(outs)} output planes")
        planes = (C.c_void_p * len(outs))(*[o.ptr for o in outs])
        _lib.check(_lib.lib().topo_amd_tpi_multi_dev(*self._head(), int(sz.size), sz.ctypes.data_as(_lib._i32p), o0, on,
                                                     planes), "tpi_multi_dev")

    def gaussian(self, sigma_y, sigma_x, out, out_row0=None, out_rows=None):
        o0, on = self._range(out_row0, out_rows)
        _lib.check(_lib.lib().topo_amd_gaussian_dev(*self._head(), float(sigma_y), float(sigma_x),
                                                    o0, on, out.ptr), "gaussian_dev")

    def gradient(self, sigma, res_x, res_y, sig_ratio=1.0, dx=None, dy=None, slope=None,
                 aspect=None, out_row0=None, out_rows=None):
        o0, on = self._range(out_row0, out_rows)
        rx = np.ascontiguousarray(res_x, dtype=np.float64)
        ry = np.ascontiguousarray(res_y, dtype=np.float64)
        mode = _lib.RES_SCALAR if rx.size == 1 and ry.size == 1 else _lib.RES_1D
        if mode == _lib.RES_1D:
            assert rx.size == self.nx and ry.size == self.gny
        p = [a.ptr if a else None for a in (dx, dy, slope, aspect)]
        _lib.check(_lib.lib().topo_amd_gradient_dev(*self._head(), float(sigma), float(sig_ratio),
                                                    mode, _lib.ptr(rx), _lib.ptr(ry), o0, on, *p),
                   "gradient_dev")

    def sx(self, dj, di, dist, window, height, out, out_row0=None, out_rows=None):
        o0, on = self._range(out_row0, out_rows)
        dj = np.ascontiguousarray(dj, dtype=np.int32)
        di = np.ascontiguousarray(di, dtype=np.int32)
        dist = np.ascontiguousarray(dist, dtype=np.float64)
        _lib.check(_lib.lib().topo_amd_sx_dev(*self._head(), dj.ctypes.data_as(_lib._i32p),
                                              di.ctypes.data_as(_lib._i32p),
                                              dist.ctypes.data_as(_lib._f64p), dist.size, int(window),
                                              float(height), o0, on, out.ptr), "sx_dev")

    def sx_multi(self, sectors, height, outs, out_row0=None, out_rows=None):
        """Sx of several azimuth sectors in one pass.  sectors: [(window, dj, di, dist), ...] as
        returned by ``sx_offsets``; outs: one DeviceArray per sector.  Same bits as ``sx`` per sector."""
        o0, on = self._range(out_row0, out_rows)
        first, dj, di, dist, window = pack_sectors(sectors)
        planes = (C.c_void_p * len(outs))(*[o.ptr for o in outs])
        _lib.check(_lib.lib().topo_amd_sx_multi_dev(
            *self._head(), len(sectors), first.ctypes.data_as(_lib._i32p), dj.ctypes.data_as(_lib._i32p),
            di.ctypes.data_as(_lib._i32p), dist.ctypes.data_as(_lib._f64p), window.ctypes.data_as(_lib._i32p),
            float(height), o0, on, planes), "sx_multi_dev")

    def valley_ridge(self, taps, ksize, angles, n_planes, mean, stdev, norm, direction, out_row0=None,
                     out_rows=None):
        """taps / ksize / angles as returned by ``topo._valley_ridge_tables``; mean / stdev of the
        WHOLE DEM (``mean_std`` for a device-resident one)."""
        o0, on = self._range(out_row0, out_rows)
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        ksize = np.ascontiguousarray(ksize, dtype=np.int32)
        angles = np.ascontiguousarray(angles, dtype=np.float32)
        _lib.check(_lib.lib().topo_amd_valley_ridge_dev(
            *self._head(), taps.ctypes.data_as(_lib._vp), ksize.ctypes.data_as(_lib._i32p),
            angles.ctypes.data_as(_lib._vp), ksize.size, int(n_planes), float(mean), float(stdev), o0, on,
            norm.ptr, direction.ptr), "valley_ridge_dev")


def mean_std(array):
    """(mean, population std) of a DeviceArray, accumulated in float64 on the GPU."""
    m, s = C.c_double(), C.c_double()
    _lib.check(_lib.lib().topo_amd_mean_std_dev(array.ptr, array.rows * array.nx, C.byref(m), C.byref(s)),
               "mean_std_dev")
    return m.value, s.value


def pack_sectors(sectors):
    """Concatenated tables of the multi-azimuth entry points: (first, dj, di, dist, window)."""
    if len(sectors) == 0:
        raise ValueError("sx_multi needs at least one sector")
    counts = [len(np.atleast_1d(sec[1])) for sec in sectors]
    first = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    dj = np.ascontiguousarray(np.concatenate([np.atleast_1d(sec[1]) for sec in sectors]), dtype=np.int32)
    di = np.ascontiguousarray(np.concatenate([np.atleast_1d(sec[2]) for sec in sectors]), dtype=np.int32)
    dist = np.ascontiguousarray(np.concatenate([np.atleast_1d(sec[3]) for sec in sectors]), dtype=np.float64)
    window = np.array([int(sec[0]) for sec in sectors], dtype=np.int32)
    return first, dj, di, dist, window


def sx_offsets(azimuth, radius, dx, dy, azimuth_arc=10.0, azimuth_steps=15, radius_min=0.0):
    """(window, dj, di, dist) for the C ABI from Sx parameters and mean grid spacing."""
    from . import topo  # noqa: PLC0415
    return topo._sx_sector(azimuth, radius, dx, dy, azimuth_arc, azimuth_steps, radius_min)
