"""The pipelined host-buffer path - what ``topo.tpi(ndarray)`` runs at real sizes (reference call shape: README.md:90,
topo.py:138) - pinned against the device-resident call and the oracle.

``run_pipelined`` (csrc/capi.hip) only cuts a call into row chunks when the array holds three chunks or more; with the
default 64 MB chunks no array of a test does.  Here ``TOPO_AMD_HOST_CHUNK_MB=1`` makes chunks of 960 rows (the minimum: whole
tile rows of every kernel) and the DEMs have 3100 rows, so every call below runs in THREE chunks (960, 960, 1180 rows) - upload
stream, compute stream, download stream, 2 x 3 events, the block views "the rows uploaded so far", a downloader thread or not - and
``topo_amd_host_chunks`` proves it.  Every ``*_f32`` entry point x {pageable, page-locked arrays} x downloads issued by
{default rule, a second thread, the calling thread} must give the bits of the serial order (``TOPO_AMD_HOST_PIPELINE=0``) and
of the ``*_dev`` call on the same data; the oracle's tolerances (tests/test_gpu_parity.py) hold on the result.  Rasters: whole
metres, fractional elevations, millimetres, and one with a NaN row and a -9999 strip lying across a chunk seam (row 960).
The environment switches are read at every call, so one process walks through them."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

from oracle import topo_oracle as orc

pytestmark = pytest.mark.gpu

from topo_descriptors_amd import _lib, device as d, topo  # noqa: E402

NY, NX = 3100, 1024
SEAM = 960  # first row of the second chunk
ENV = ("TOPO_AMD_HOST_CHUNK_MB", "TOPO_AMD_HOST_PIPELINE", "TOPO_AMD_HOST_DOWNLOADS")
# (pipeline switch, downloads) - the first entry is the reference order: one chunk
MODES = [("0", None), (None, None), (None, "thread"), (None, "inline")]


def rasters():
    base_i = orc.synthetic_dem(NY, NX, seed=11, integer=True)
    base_f = orc.synthetic_dem(NY, NX, seed=12, integer=False)
    out = {"metres": base_i, "fractional": base_f, "mm": (base_f * 1000.0).astype(np.float32)}
    a = base_f.copy()
    a[SEAM - 2, :] = np.nan                           # a NaN row just above the seam: its footprint crosses it
    a[SEAM - 9: SEAM + 14, 300:700] = -9999.0          # nodata lying across the seam
    a[2 * SEAM - 1: 2 * SEAM + 1, 40:60] = np.inf      # and a patch of inf on the next seam
    out["nan_row+nodata_at_the_seam"] = a
    return out


RASTERS = rasters()


class Pinned:
    """A page-locked float32 array from topo_amd_host_alloc."""

    def __init__(self, shape):
        self.p = C.c_void_p()
        n = int(np.prod(shape)) * 4
        _lib.check(_lib.lib().topo_amd_host_alloc(C.byref(self.p), n), "host_alloc")
        self.a = np.frombuffer((C.c_char * n).from_address(self.p.value), dtype=np.float32).reshape(shape)

    def free(self):
        self.a = None
        _lib.check(_lib.lib().topo_amd_host_free(self.p), "host_free")


class Arrays:
    """Input and n_out output planes, pageable or page-locked."""

    def __init__(self, dem, n_out, pinned):
        self.pins = []
        if pinned:
            self.pins = [Pinned(dem.shape) for _ in range(n_out + 1)]
            self.src = self.pins[0].a
            self.src[:] = dem
            self.outs = [p.a for p in self.pins[1:]]
        else:
            self.src = np.ascontiguousarray(dem)
            self.outs = [np.empty_like(dem) for _ in range(n_out)]
        for o in self.outs:
            o[:] = -12345.0  # a row no download reaches would show

    def free(self):
        for p in self.pins:
            p.free()


def set_mode(pipeline, downloads):
    os.environ["TOPO_AMD_HOST_CHUNK_MB"] = "1"
    for key, val in (("TOPO_AMD_HOST_PIPELINE", pipeline), ("TOPO_AMD_HOST_DOWNLOADS", downloads)):
        if val is None:
            os.environ.pop(key, None)
        else:
            os.environ[key] = val


@pytest.fixture(autouse=True)
def clean_env():
    saved = {k: os.environ.get(k) for k in ENV}
    yield
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def crc(planes):
    return [zlib.crc32(np.ascontiguousarray(p).view(np.uint8)) for p in planes]


def f32p(a):
    return a.ctypes.data_as(_lib._f32p)


SECTORS = [d.sx_offsets(a, 500.0, 30.0, -30.0) for a in (350.0, 0.0, 45.0)]
X = 2600000.0 + 30.0 * np.arange(NX)
Y = 1200000.0 - 30.0 * np.arange(NY)
RES = orc.grid_resolution(X, Y)


def valley_tables():
    kernels = topo._valley_kernels(7, [0, 0.15, 0.3])
    return topo._valley_ridge_tables(kernels, np.arange(0, 180, 45, dtype=np.float32)) + (kernels.shape[0],)


# name -> (number of output planes, host call (src, outs), device call (Block, device outs), allow EEMPTY)
def host_tpi(size, sigma=0.0):
    return lambda lib, s, o: lib.topo_amd_tpi_f32(f32p(s), NY, NX, size, sigma, f32p(o[0]))


def host_std(size):
    return lambda lib, s, o: lib.topo_amd_std_f32(f32p(s), NY, NX, size, 0.0, f32p(o[0]))


def host_tpi_std(size, sigma=0.0):
    return lambda lib, s, o: lib.topo_amd_tpi_std_f32(f32p(s), NY, NX, size, sigma, f32p(o[0]), f32p(o[1]))


def host_multi(sizes):
    sz = np.asarray(sizes, dtype=np.int32)
    sg = np.zeros(len(sizes))

    def call(lib, s, o):
        n = len(sizes)
        t = (C.c_void_p * n)(*[a.ctypes.data for a in o[:n]])
        sd = (C.c_void_p * n)(*[a.ctypes.data for a in o[n:]])
        return lib.topo_amd_tpi_std_multi_f32(f32p(s), NY, NX, n, sz.ctypes.data_as(_lib._i32p), sg.ctypes.data_as(_lib._f64p), t, sd)
    return call


def host_gradient(sigma, one_d=False):
    if one_d:
        rx = np.ascontiguousarray(np.broadcast_to(RES["x"], (NX,)), dtype=np.float64)
        ry = np.ascontiguousarray(np.broadcast_to(RES["y"], (NY,)), dtype=np.float64)
        mode = _lib.RES_1D
    else:
        rx, ry, mode = np.array([30.0]), np.array([-30.0]), _lib.RES_SCALAR
    return lambda lib, s, o: lib.topo_amd_gradient_f32(f32p(s), NY, NX, sigma, 1.0, mode, _lib.ptr(rx), _lib.ptr(ry), *[f32p(a) for a in o])


def host_sx(sector):
    w, dj, di, dist = sector
    dj, di = np.ascontiguousarray(dj, dtype=np.int32), np.ascontiguousarray(di, dtype=np.int32)
    dist = np.ascontiguousarray(dist, dtype=np.float64)
    return lambda lib, s, o: lib.topo_amd_sx_f32(f32p(s), NY, NX, dj.ctypes.data_as(_lib._i32p), di.ctypes.data_as(_lib._i32p),
                                                 dist.ctypes.data_as(_lib._f64p), dist.size, int(w), 10.0, f32p(o[0]))


def host_sx_multi(sectors):
    first, dj, di, dist, window = d.pack_sectors(sectors)

    def call(lib, s, o):
        planes = (C.c_void_p * len(sectors))(*[a.ctypes.data for a in o])
        return lib.topo_amd_sx_multi_f32(f32p(s), NY, NX, len(sectors), first.ctypes.data_as(_lib._i32p), dj.ctypes.data_as(_lib._i32p),
                                         di.ctypes.data_as(_lib._i32p), dist.ctypes.data_as(_lib._f64p),
                                         window.ctypes.data_as(_lib._i32p), 10.0, planes)
    return call


def dev_tpi_std(size, want_t, want_s):
    def call(blk, o):
        blk.tpi_std(size, tpi=o[0] if want_t else None, std=o[-1] if want_s else None)
    return call


def dev_multi(sizes):
    def call(blk, o):
        for k, size in enumerate(sizes):
            blk.tpi_std(size, tpi=o[k], std=o[len(sizes) + k])
    return call


CASES = {
    "tpi7": (1, host_tpi(7), dev_tpi_std(7, True, False)),
    "tpi31": (1, host_tpi(31), dev_tpi_std(31, True, False)),
    "tpi67": (1, host_tpi(67), dev_tpi_std(67, True, False)),
    "tpi6_even": (1, host_tpi(6), dev_tpi_std(6, True, False)),
    "std7": (1, host_std(7), dev_tpi_std(7, False, True)),
    "std67": (1, host_std(67), dev_tpi_std(67, False, True)),
    "tpi_std21": (2, host_tpi_std(21), dev_tpi_std(21, True, True)),
    "tpi_std67": (2, host_tpi_std(67), dev_tpi_std(67, True, True)),
    "tpi_std_multi": (6, host_multi([5, 9, 67]), dev_multi([5, 9, 67])),
    "gauss3.25": (1, lambda lib, s, o: lib.topo_amd_gauss_f32(f32p(s), NY, NX, 3.25, 3.25, f32p(o[0])),
                  lambda blk, o: blk.gaussian(3.25, 3.25, o[0])),
    "gauss30.25": (1, lambda lib, s, o: lib.topo_amd_gauss_f32(f32p(s), NY, NX, 30.25, 30.25, f32p(o[0])),
                   lambda blk, o: blk.gaussian(30.25, 30.25, o[0])),
    "gradient3.25": (4, host_gradient(3.25), lambda blk, o: blk.gradient(3.25, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3])),
    "gradient30.25_res1d": (4, host_gradient(30.25, True),
                            lambda blk, o: blk.gradient(30.25, RES["x"] * np.ones(NX), RES["y"] * np.ones(NY), dx=o[0], dy=o[1], slope=o[2], aspect=o[3])),
    "sobel": (2, lambda lib, s, o: lib.topo_amd_sobel_f32(f32p(s), NY, NX, f32p(o[0]), f32p(o[1])), None),
    "sx": (1, host_sx(SECTORS[1]), lambda blk, o: blk.sx(SECTORS[1][1], SECTORS[1][2], SECTORS[1][3], SECTORS[1][0], 10.0, o[0])),
    "sx_multi": (3, host_sx_multi(SECTORS), lambda blk, o: blk.sx_multi(SECTORS, 10.0, o)),
}


def sobel_dev(blk, o):
    _lib.check(_lib.lib().topo_amd_sobel_dev(*blk._head(), 0, NY, o[0].ptr, o[1].ptr), "sobel_dev")


CASES["sobel"] = (2, CASES["sobel"][1], sobel_dev)


def device_reference(dem, n_out, dev_call):
    dev = d.DeviceArray.from_host(dem)
    outs = [d.DeviceArray(NY, NX) for _ in range(n_out)]
    dev_call(d.Block(dev), outs)
    d.sync()
    host = [o.to_host() for o in outs]
    for a in outs + [dev]:
        a.free()
    return host


def check_oracle(case, raster, dem, planes):
    """The tolerances of tests/test_gpu_parity.py on the pipelined result (whole metres and fractional elevations)."""
    def rel(got, want):
        ok = np.isfinite(want)
        return float(np.max(np.abs(got[ok] - want[ok])) / max(float(np.max(np.abs(want[ok]))), 1e-30))
    if case in ("tpi7", "tpi31", "tpi67", "tpi6_even"):
        size = {"tpi7": 7, "tpi31": 31, "tpi67": 67, "tpi6_even": 6}[case]
        want = orc.tpi_exact(dem, size)
        bound = 2.5e-4 + (2.0 ** -9 if raster == "fractional" and size >= 19 else 0.0)  # (the scaled route: tpi_alone_bound)
        assert float(np.max(np.abs(planes[0] - want))) <= bound, (case, raster)
    elif case in ("std7", "std67"):
        want = orc.std_exact(dem, 7 if case == "std7" else 67)
        assert rel(planes[0].astype(np.float64), want) <= 1e-4, (case, raster)
    elif case in ("tpi_std21", "tpi_std67"):
        size = 21 if case == "tpi_std21" else 67
        assert float(np.max(np.abs(planes[0] - orc.tpi_exact(dem, size)))) <= 2.5e-4, (case, raster)
        assert rel(planes[1].astype(np.float64), orc.std_exact(dem, size)) <= 1e-4, (case, raster)
    elif case.startswith("gauss"):
        sigma = float(case[5:])
        want = orc.gaussian_exact(dem, sigma)
        assert float(np.max(np.abs(planes[0] - want))) <= 1e-3, (case, raster)  # metres, against the float64 filter
    elif case.startswith("gradient"):
        sigma = 3.25 if case == "gradient3.25" else 30.25
        want = orc.gradient_exact(dem, sigma, RES)
        for k in (0, 1, 2):
            assert rel(planes[k], want[k]) <= 1e-4, (case, raster, k)
        steep = want[2] > 0.1
        diff = orc.wrapped_angle_diff(planes[3][steep], want[3][steep])
        # (the conditioning-aware bound of tests/test_gpu_blocks.py: 1e-4 x 360 plus what an error of 1e-4 of the largest
        # derivative turns the direction of a gentle gradient by)
        gmag = np.hypot(want[0], want[1])[steep]
        tol = 1e-4 * max(float(np.max(np.abs(want[0]))), float(np.max(np.abs(want[1]))))
        assert np.all(diff <= 0.036 + np.degrees(np.arctan(np.sqrt(2.0) * tol / gmag))), (case, raster)
    elif case == "sobel":
        want = orc.sobel_exact(dem)
        for k in (0, 1):
            assert rel(planes[k], want[k]) <= 1e-6, (case, raster, k)
    elif case in ("sx", "sx_multi"):
        # the oracle on a band of rows around the first chunk seam (its frame rows excluded)
        lo, hi = SEAM - 90, SEAM + 90
        secs = [SECTORS[1]] if case == "sx" else SECTORS
        for plane, (w, dj, di, dist) in zip(planes, secs):
            band = orc.sx_rolling(dem[lo:hi, :256], w, np.stack([dj, di], axis=1), dist, 10.0)
            got = plane[lo + w: hi - w, w: 256 - w]
            assert float(np.max(np.abs(got - band[w:-w, w:-w]))) <= 1e-4 * 90.0, (case, raster)


@pytest.mark.parametrize("raster", sorted(RASTERS))
@pytest.mark.parametrize("case", sorted(CASES))
def test_pipelined_host_call_has_the_serial_and_the_device_bits(case, raster):
    n_out, host_call, dev_call = CASES[case]
    dem = RASTERS[raster]
    lib = _lib.lib()
    want = device_reference(dem, n_out, dev_call)
    want_crc = crc(want)
    for pinned in (False, True):
        arrays = Arrays(dem, n_out, pinned)
        try:
            for pipeline, downloads in MODES:
                for o in arrays.outs:
                    o[:] = -12345.0
                set_mode(pipeline, downloads)
                rc = host_call(lib, arrays.src, arrays.outs)
                assert rc == 0, (case, raster, rc, lib.topo_amd_last_error())
                chunks = d.host_chunks()
                if pipeline == "0":
                    assert chunks == 1, (case, chunks)
                else:
                    assert chunks == 3, (case, raster, pinned, downloads, chunks)  # 3100 rows: 960 + 960 + 1180
                got_crc = crc(arrays.outs)
                if got_crc != want_crc:
                    bad = [(k, int((~((g == w) | (np.isnan(g) & np.isnan(w)))).sum())) for k, (g, w) in enumerate(zip(arrays.outs, want))]
                    raise AssertionError((case, raster, "pinned" if pinned else "pageable", pipeline, downloads, bad))
        finally:
            arrays.free()
    if raster in ("metres", "fractional"):
        check_oracle(case, raster, dem, want)


def test_valley_ridge_host_call_is_serial_and_has_the_device_bits():
    """topo_amd_valley_ridge_f32 does not go through the pipeline (0.5 s of kernels per 20 ms of copies at full size, and its
    FFT route is not cut-invariant): one chunk whatever the switches say, the device call's bits."""
    dem = RASTERS["metres"][:, :512].copy()
    ny, nx = dem.shape
    taps, ksize, angles, n_planes = valley_tables()
    taps = np.ascontiguousarray(taps, dtype=np.float32)
    ksize = np.ascontiguousarray(ksize, dtype=np.int32)
    angles = np.ascontiguousarray(angles, dtype=np.float32)
    mean, stdev = float(dem.astype(np.float64).mean()), float(dem.astype(np.float64).std())
    dev = d.DeviceArray.from_host(dem)
    n_dev, d_dev = d.DeviceArray(ny, nx), d.DeviceArray(ny, nx)
    d.Block(dev).valley_ridge(taps, ksize, angles, n_planes, mean, stdev, n_dev, d_dev)
    d.sync()
    want = [n_dev.to_host(), d_dev.to_host()]
    set_mode(None, None)
    norm, direction = np.empty_like(dem), np.empty_like(dem)
    _lib.check(_lib.lib().topo_amd_valley_ridge_f32(f32p(dem), ny, nx, taps.ctypes.data_as(_lib._vp), ksize.ctypes.data_as(_lib._i32p),
                                                    angles.ctypes.data_as(_lib._vp), ksize.size, int(n_planes), mean, stdev,
                                                    f32p(norm), f32p(direction)), "valley_ridge_f32")
    assert d.host_chunks() == 1
    assert np.array_equal(norm, want[0]) and np.array_equal(direction, want[1])
    for a in (dev, n_dev, d_dev):
        a.free()


def test_topo_tpi_of_an_ndarray_takes_the_pipeline():
    """The literal drop-in call (reference README.md:90): ``topo.tpi(ndarray, size)`` of an array of three chunks and more."""
    set_mode(None, None)
    dem = RASTERS["metres"]
    got = topo.tpi(dem, 67)
    assert d.host_chunks() == 3
    set_mode("0", None)
    assert np.array_equal(topo.tpi(dem, 67), got)
    assert d.host_chunks() == 1
    t, s = topo.tpi_std(dem, 67)
    assert np.array_equal(t, got)


def test_many_chunks_of_a_tall_array():
    """A 6200-row array in six chunks (five of 960 rows and one of 1400), with filters whose reach (33 / 122 / 17 rows) is well
    below a chunk, nodata across the first cut and a NaN row next to the last one."""
    ny, nx = 6200, 512
    dem = orc.synthetic_dem(ny, nx, seed=21, integer=False)
    dem[955:966, 100:300] = -9999.0
    dem[4799, :] = np.nan
    lib = _lib.lib()
    dev = d.DeviceArray.from_host(dem)
    blk = d.Block(dev)
    t, s4 = d.DeviceArray(ny, nx), [d.DeviceArray(ny, nx) for _ in range(4)]
    x = d.DeviceArray(ny, nx)
    blk.tpi_std(67, tpi=t)
    blk.gradient(30.25, [30.0], [-30.0], dx=s4[0], dy=s4[1], slope=s4[2], aspect=s4[3])
    blk.sx(SECTORS[1][1], SECTORS[1][2], SECTORS[1][3], SECTORS[1][0], 10.0, x)
    d.sync()
    want_t, want_g, want_x = t.to_host(), [a.to_host() for a in s4], x.to_host()
    for a in [dev, t, x] + s4:
        a.free()
    w, dj, di, dist = SECTORS[1]
    dj, di = np.ascontiguousarray(dj, dtype=np.int32), np.ascontiguousarray(di, dtype=np.int32)
    dist = np.ascontiguousarray(dist, dtype=np.float64)
    rx, ry = np.array([30.0]), np.array([-30.0])
    set_mode(None, None)
    out = np.full_like(dem, -12345.0)
    assert lib.topo_amd_tpi_f32(f32p(dem), ny, nx, 67, 0.0, f32p(out)) == 0
    assert d.host_chunks() == 6, d.host_chunks()
    assert np.array_equal(out, want_t, equal_nan=True)
    outs = [np.full_like(dem, -12345.0) for _ in range(4)]
    assert lib.topo_amd_gradient_f32(f32p(dem), ny, nx, 30.25, 1.0, _lib.RES_SCALAR, _lib.ptr(rx), _lib.ptr(ry), *[f32p(o) for o in outs]) == 0
    for g, wg in zip(outs, want_g):
        assert np.array_equal(g, wg, equal_nan=True)
    out[:] = -12345.0
    assert lib.topo_amd_sx_f32(f32p(dem), ny, nx, dj.ctypes.data_as(_lib._i32p), di.ctypes.data_as(_lib._i32p),
                               dist.ctypes.data_as(_lib._f64p), dist.size, int(w), 10.0, f32p(out)) == 0
    assert np.array_equal(out, want_x, equal_nan=True)
