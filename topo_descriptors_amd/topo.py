"""``topo.*`` descriptor functions with the reference's signatures, on MI355X.

Same names, argument meaning, return types and errors as the reference's
``topo_descriptors/topo.py`` (tpi :145, std :273, dem :62, circular_kernel :191, gradient :598,
sobel :658, sx :776 and its geometry helpers :861-925).  The per-pixel arithmetic runs in
hand-written HIP kernels behind the C ABI of ``libtopo_amd.so``; this module only prepares
parameters, hands over C-contiguous float32 buffers and wraps the results.
"""
import ctypes as C
import logging
import os

import numpy as np

from . import _lib, helpers as hlp

logger = logging.getLogger(__name__)


def __getattr__(name):
    """The reference keeps its batch wrappers in this module (``tp.compute_tpi(...)``, topo.py:24-141 and
    on); here they live in ``batch`` (which imports this module) and are resolved on first use."""
    if name.startswith("compute_"):
        from . import batch  # noqa: PLC0415
        if hasattr(batch, name):
            return getattr(batch, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


# ---- small utilities ------------------------------------------------------------------------
def _unwrap(dem):
    """ndarray view of an ndarray or DataArray-like input, plus a re-wrapper."""
    if isinstance(dem, np.ndarray):
        return dem, lambda out: out
    if hasattr(dem, "values") and hasattr(dem, "copy"):
        def rewrap(out, src=dem):
            try:
                return src.copy(data=out)
            except TypeError:
                return out
        return np.asarray(dem.values), rewrap
    return np.asarray(dem), lambda out: out


def _check_2d(a, who):
    if a.ndim != 2 or a.shape[0] < 1 or a.shape[1] < 1:
        raise ValueError(f"{who}: expected a non-empty 2-D array, got shape {a.shape}")


def _sigma_arg(sigma):
    return float(sigma) if sigma else 0.0


# ---- disc kernel, TPI, STD ---------------------------------------------------------------------
def circular_kernel(size):
    """0/1 float32 disc of diameter ``size``; a full square below 5 (reference topo.py:191)."""
    size = int(size)
    mask = np.empty((size, size), dtype=np.float32)
    if size == 0:
        return mask
    lib = _lib.load()  # pure host geometry: no GPU needed
    _lib.check(lib.topo_amd_disc_mask(size, mask.ctypes.data_as(_lib._f32p)), "topo_amd_disc_mask")
    return mask


def tpi(dem, size, sigma=None):
    """Topographic position index: elevation minus the mean of the disc neighbourhood of
    diameter ``size`` pixels, centre excluded; optional Gaussian pre-smoothing ``sigma``
    (reference topo.py:145-181).  float32 in, float32 out; DataArray in, DataArray out.

    With ``sigma`` the pre-smoothing is :func:`dem`'s: see there for how far a non-finite sample reaches.  Without it a
    sample that is not finite (or beyond +-2**24) is MISSING, and exactly the pixels whose disc holds one are NaN - the
    footprint of the disc, whatever the tiles or row blocks; the reference's FFT convolution makes the whole array NaN.

    Accuracy.  On a DEM of whole metres the result is the float64 evaluation of the reference's formula, rounded to
    float32.  Where a disc of 19 pixels or more holds fractional elevations, the neighbourhood sum is taken on ``x``
    in units of 2**-k m (one integer chain instead of two; k = 8 for an ordinary DEM, up to 16 for a raster whose whole
    value range is small): each sample is off by at most 2**-9 m = 1.95 mm, hence so are
    the mean and TPI (about 0.02 mm rms at 67 pixels when the fractional parts are spread evenly) - the order of the
    1.4 - 1.7 mm the reference's own float32 FFT is off by, and far inside 1e-4 of the range.  Smaller discs, and
    :func:`tpi_std`, are exact to 2**-16 m; ``TOPO_AMD_TPI_FRACTION_EXACT=1`` makes this function so as well, at about
    twice the time on such a DEM."""
    values, rewrap = _unwrap(dem)
    _check_2d(values, "tpi")
    src = _lib.as_f32(values)
    out = np.empty_like(src)
    lib = _lib.lib()
    _lib.check(lib.topo_amd_tpi_f32(_lib.ptr(src), src.shape[0], src.shape[1], int(size),
                                    _sigma_arg(sigma), _lib.ptr(out)), "topo_amd_tpi_f32")
    return rewrap(out)


def std(dem, size, sigma=None):
    """Sample standard deviation inside the disc window, with the reference's int32
    truncation of the squared term (reference topo.py:273-307).  Returns float64.  Non-finite samples: as for
    :func:`tpi` (and :func:`dem` when ``sigma`` is given)."""
    values, _ = _unwrap(dem)
    _check_2d(values, "std")
    src = _lib.as_f32(values)
    out = np.empty_like(src)
    lib = _lib.lib()
    _lib.check(lib.topo_amd_std_f32(_lib.ptr(src), src.shape[0], src.shape[1], int(size),
                                    _sigma_arg(sigma), _lib.ptr(out)), "topo_amd_std_f32")
    return out.astype(np.float64)


def tpi_std(dem, size, sigma=None):
    """TPI and STD of the same window in one pass over the DEM (no reference counterpart:
    the reference convolves three times for the pair)."""
    values, _ = _unwrap(dem)
    _check_2d(values, "tpi_std")
    src = _lib.as_f32(values)
    t = np.empty_like(src)
    s = np.empty_like(src)
    lib = _lib.lib()
    _lib.check(lib.topo_amd_tpi_std_f32(_lib.ptr(src), src.shape[0], src.shape[1], int(size),
                                        _sigma_arg(sigma), _lib.ptr(t), _lib.ptr(s)),
               "topo_amd_tpi_std_f32")
    return t, s.astype(np.float64)


def tpi_std_multi(dem, sizes, sigmas=None, want_tpi=True, want_std=True):
    """TPI and / or STD for several disc sizes from one upload of the DEM (the multi-scale half):
    what the reference's ``compute_tpi`` / ``compute_std`` loops do scale by scale (topo.py:88-141, :216-269).
    Returns ``(tpis, stds)``, lists of planes in the order of ``sizes`` (``None`` for the kind not asked
    for); every plane has the bits of the single call."""
    import ctypes as C
    values, _ = _unwrap(dem)
    _check_2d(values, "tpi_std_multi")
    src = _lib.as_f32(values)
    sizes = np.ascontiguousarray(np.atleast_1d(sizes), dtype=np.int32)
    n = int(sizes.size)
    if sigmas is None:
        sig = np.zeros(n, dtype=np.float64)
    else:
        sig = np.array([_sigma_arg(v) for v in np.broadcast_to(np.asarray(sigmas, dtype=object), (n,))],
                       dtype=np.float64)
    tpis = [np.empty_like(src) for _ in range(n)] if want_tpi else None
    stds = [np.empty_like(src) for _ in range(n)] if want_std else None

    def plane_list(planes):
        if planes is None:
            return None
        return (C.c_void_p * n)(*[p.ctypes.data for p in planes])

    lib = _lib.lib()
    t_arg, s_arg = plane_list(tpis), plane_list(stds)
    _lib.check(lib.topo_amd_tpi_std_multi_f32(_lib.ptr(src), src.shape[0], src.shape[1], n,
                                              sizes.ctypes.data_as(C.POINTER(C.c_int32)),
                                              sig.ctypes.data_as(C.POINTER(C.c_double)), t_arg, s_arg),
               "topo_amd_tpi_std_multi_f32")
    return tpis, ([s.astype(np.float64) for s in stds] if stds is not None else None)


# ---- Gaussian, Sobel, gradient ------------------------------------------------------------------
def dem(dem, sigma):
    """Gaussian-smoothed DEM, reflect boundary, 4-sigma truncation (reference topo.py:62-80).
    ``sigma`` may be a scalar or an (axis0, axis1) pair.

    Non-finite samples: a NaN / inf in the DEM makes non-finite every output whose window contains it, as in
    ``scipy.ndimage.gaussian_filter`` - and exactly those when the smoothing runs on the matrix cores (Gaussian radius
    ``int(4 sigma + 0.5)`` of 4 ... 121, any DEM width from 4 columns): those kernels keep such samples
    (and absurd ones, ``|x| > 1e5``) out of the matrix pipe, mark the tiles whose windows hold one, and a repair pass
    recomputes in float32, over each output's own window, exactly the outputs that see one.  (A raster whose ordinary
    values lie beyond 1e5 - a DEM in millimetres - is recognised by sampling at the first call and smoothed by the
    vector-ALU kernels instead.)  The vector-ALU kernels
    (shorter or much longer filters) pad their taps to chunks of 8 or 16 and spoil up to one chunk more
    towards lower indices.  ``tests/test_gpu_parity.py::test_gaussian_nan_footprint`` pins both.
    """
    values, rewrap = _unwrap(dem)
    _check_2d(values, "dem")
    sig = np.broadcast_to(np.asarray(sigma, dtype=np.float64), (2,))
    src = _lib.as_f32(values)
    out = np.empty_like(src)
    lib = _lib.lib()
    _lib.check(lib.topo_amd_gauss_f32(_lib.ptr(src), src.shape[0], src.shape[1], float(sig[0]),
                                      float(sig[1]), _lib.ptr(out)), "topo_amd_gauss_f32")
    return rewrap(out)


def sobel(dem):
    """Sobel derivative pair (dx, dy), kernel / 8, reflect boundary (reference topo.py:658-685)."""
    values, _ = _unwrap(dem)
    _check_2d(values, "sobel")
    src = _lib.as_f32(values)
    dx = np.empty_like(src)
    dy = np.empty_like(src)
    lib = _lib.lib()
    _lib.check(lib.topo_amd_sobel_f32(_lib.ptr(src), src.shape[0], src.shape[1], _lib.ptr(dx),
                                      _lib.ptr(dy)), "topo_amd_sobel_f32")
    return dx, dy


def _resolution_args(res_meters, shape):
    """(mode, x array, y array) for the C ABI from the reference's res_meters dict."""
    rx = np.asarray(res_meters["x"], dtype=np.float64)
    ry = np.asarray(res_meters["y"], dtype=np.float64)
    ny, nx = shape
    if rx.ndim == 0 and ry.ndim == 0:
        return _lib.RES_SCALAR, np.ascontiguousarray(rx.reshape(1)), np.ascontiguousarray(ry.reshape(1))
    if rx.ndim <= 1 and ry.ndim <= 1:
        rx = np.ascontiguousarray(np.broadcast_to(rx, (nx,)))
        ry = np.ascontiguousarray(np.broadcast_to(ry, (ny,)))
        return _lib.RES_1D, rx, ry
    # per-pixel resolutions (WGS84 grids): y may still be a column vector
    if ry.ndim == 1:
        ry = ry[:, None]
    rx = np.ascontiguousarray(np.broadcast_to(rx, shape), dtype=np.float32)
    ry = np.ascontiguousarray(np.broadcast_to(ry, shape), dtype=np.float32)
    return _lib.RES_2D, rx, ry


def gradient(dem, sigma, res_meters, sig_ratio=1):
    """[dx, dy, slope, aspect] of the Gaussian-smoothed DEM (reference topo.py:598-644).

    ``sigma <= 1`` uses the Sobel pair; ``sig_ratio != 1`` smooths with ``sigma*sig_ratio``
    perpendicular to each derivative.  Derivatives are divided by the signed grid
    resolution ``res_meters`` (second return of :func:`helpers.scale_to_pixel`); slope in
    degrees; aspect in [0, 360) with north-facing = 0 and east-facing = 90.  Non-finite samples reach as far as in
    :func:`dem` (plus the one pixel of the finite difference)."""
    values, _ = _unwrap(dem)
    _check_2d(values, "gradient")
    src = _lib.as_f32(values)
    mode, rx, ry = _resolution_args(res_meters, src.shape)
    outs = [np.empty_like(src) for _ in range(4)]
    lib = _lib.lib()
    _lib.check(lib.topo_amd_gradient_f32(_lib.ptr(src), src.shape[0], src.shape[1], float(sigma),
                                         float(sig_ratio), mode, _lib.ptr(rx), _lib.ptr(ry),
                                         *[_lib.ptr(o) for o in outs]), "topo_amd_gradient_f32")
    return outs


# ---- Sx ---------------------------------------------------------------------------------------------
def _sx_distance(radius, dx, dy):
    """Distance in metres from the centre of the Sx search window to each of its cells
    (reference topo.py:861-878).  The window has ceil(2*radius_px + 1) cells per side."""
    radius_px = max(radius / np.abs(dy), radius / np.abs(dx))
    side = 2 * radius_px + 1
    mid = np.floor(side / 2)
    steps = np.arange(side) - mid
    return np.hypot((steps * dy)[:, None], (steps * dx)[None, :])


def _sx_source_idx_delta(azimuths, radius, dx, dy):
    """(row, col) index offsets of the ray origins at distance ``radius`` in the directions
    ``azimuths`` (degrees); resolutions are signed (reference topo.py:881-892)."""
    rad = np.deg2rad(np.asarray(azimuths, dtype=np.float64))
    out = np.empty((rad.size, 2), dtype=np.int64)
    out[:, 0] = np.rint(radius / dy * np.cos(rad))
    out[:, 1] = np.rint(radius / dx * np.sin(rad))
    return out


def _sx_bresenhamlines(start, end):
    """Pixels on the straight lines from each ``start`` towards ``end``, both excluded
    (reference topo.py:895-925): unit steps along each ray's dominant axis, half-to-even
    rounding, cut where the L1 distance to ``end`` stops shrinking."""
    start = np.asarray(start)
    end = np.asarray(end)
    span = end - start                                   # (n_rays, 2)
    major = np.abs(span).max(axis=1)                     # steps needed per ray
    n_steps = int(major.max()) if major.size else 0
    safe = np.where(major == 0, 1, major).astype(np.float64)
    direction = np.where((major == 0)[:, None], 0.0, span / safe[:, None])
    t = np.arange(1, n_steps + 1, dtype=np.float64)
    pts = np.rint(start[:, None, :] + direction[:, None, :] * t[None, :, None]).astype(start.dtype)
    l1 = np.abs(pts - end).sum(axis=2)                   # (n_rays, n_steps)
    shrinking = np.ones_like(l1, dtype=bool)
    shrinking[:, 1:] = l1[:, 1:] <= l1[:, :-1]
    keep = shrinking & (l1 != 0)
    return pts[keep]


def _sx_scan(dem, reach, rows, cols, metres, height):
    """Kernel K6 on a host array: for every pixel the largest elevation angle (degrees) towards the ray pixels
    ``(rows[k], cols[k])`` (offsets from the pixel) at ``metres[k]`` horizontal distance, seen from ``height``
    above the pixel.  NaN distances are skipped; a frame of ``reach`` pixels stays 0."""
    dem = np.asarray(dem)
    _check_2d(dem, "_sx_rolling")
    metres = np.ascontiguousarray(metres, dtype=np.float64)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    src = _lib.as_f32(dem)
    out = np.zeros_like(src)
    ny, nx = src.shape
    if ny <= 2 * reach or nx <= 2 * reach:  # nothing but frame
        return out.astype(dem.dtype, copy=False)
    if metres.size == 0 or np.all(np.isnan(metres)):
        # nanmax over an empty / all-NaN set: NaN inside the frame, like numpy
        out[reach : ny - reach, reach : nx - reach] = np.nan
        return out.astype(dem.dtype, copy=False)
    lib = _lib.lib()
    _lib.check(lib.topo_amd_sx_f32(_lib.ptr(src), ny, nx, rows.ctypes.data_as(_lib._i32p),
                                   cols.ctypes.data_as(_lib._i32p), metres.ctypes.data_as(_lib._f64p),
                                   int(metres.size), int(reach), float(height), _lib.ptr(out)),
               "topo_amd_sx_f32")
    return out.astype(dem.dtype, copy=False)


def _sx_rolling(dem, distance, blines, height):
    """The reference's private entry (topo.py:928-953): ``distance`` is the square table of distances around
    the pixel, ``blines`` the (row, column) cells of that table the rays cross."""
    reach = int(distance.shape[0] / 2)
    cells = np.asarray(blines, dtype=np.int64).reshape(-1, 2)
    return _sx_scan(dem, reach, cells[:, 0] - reach, cells[:, 1] - reach, distance[cells[:, 0], cells[:, 1]], height)


def _sx_sector(azimuth, radius, spacing_x, spacing_y, azimuth_arc=10.0, azimuth_steps=15, radius_min=0.0):
    """The ray pixels of one sector as ``(reach, rows, cols, metres)``: offsets from the target pixel and their
    distances (NaN where closer than ``radius_min``), built from the reference's three geometry helpers so that
    the set is the reference's (topo.py:825-853)."""
    n_rays = 1 if azimuth_arc == 0 else azimuth_steps
    ray_azimuths = np.linspace(azimuth - azimuth_arc / 2, azimuth + azimuth_arc / 2, n_rays)
    table = _sx_distance(radius, spacing_x, spacing_y)
    table[table < radius_min] = np.nan
    middle = np.floor(np.array(table.shape) / 2)
    far_ends = (middle + _sx_source_idx_delta(ray_azimuths, radius, spacing_x, spacing_y)).astype(int)
    cells = np.asarray(_sx_bresenhamlines(far_ends, middle), dtype=np.int64).reshape(-1, 2)
    reach = int(table.shape[0] / 2)
    return reach, cells[:, 0] - reach, cells[:, 1] - reach, table[cells[:, 0], cells[:, 1]]


def sx(dem_ds, azimuth, radius, height=10.0, azimuth_arc=10.0, azimuth_steps=15, radius_min=0.0):
    """Sx (Winstral et al.): maximum slope towards the terrain within ``radius`` metres in a
    sector of ``azimuth_arc`` degrees around ``azimuth`` (reference topo.py:776-858).

    ``dem_ds`` must be a Dataset (TypeError otherwise, like the reference); the mean signed
    grid spacing is used; pixels closer than ``radius_min`` are ignored; ``height`` is the
    instrument height added to the target pixel."""
    if not hlp._looks_like_dataset(dem_ds):
        raise TypeError("Argument 'dem_ds' must be a xr.Dataset.")
    spacing = hlp.scale_to_pixel(radius, dem_ds)[1]
    sector = _sx_sector(azimuth, radius, spacing["x"].mean(), spacing["y"].mean(), azimuth_arc, azimuth_steps,
                        radius_min)
    return _sx_scan(hlp.get_da(dem_ds).values, *sector, height)


def sx_multi(dem_ds, azimuths, radius, height=10.0, azimuth_arc=10.0, azimuth_steps=15, radius_min=0.0):
    """``sx`` for a sequence of azimuths in one pass over the DEM: ``[sx(dem_ds, a, radius, ...) for a
    in azimuths]`` with the same bits.  The reference scans one azimuth per call (topo.py:776-858) and
    its users loop; here the DEM is uploaded once and ray pixels shared by neighbouring sectors are
    scanned once.  Give the azimuths in angular order for the sharing to apply."""
    if not hlp._looks_like_dataset(dem_ds):
        raise TypeError("Argument 'dem_ds' must be a xr.Dataset.")
    azimuths = [float(a) for a in np.atleast_1d(azimuths)]
    if len(azimuths) == 0:
        return []
    from . import device  # noqa: PLC0415
    _, res_meters = hlp.scale_to_pixel(radius, dem_ds)
    dx = res_meters["x"].mean()
    dy = res_meters["y"].mean()
    sectors = [device.sx_offsets(a, radius, dx, dy, azimuth_arc, azimuth_steps, radius_min) for a in azimuths]
    dem = np.asarray(hlp.get_da(dem_ds).values)
    _check_2d(dem, "sx_multi")
    src = _lib.as_f32(dem)
    ny, nx = src.shape
    outs = [np.zeros_like(src) for _ in sectors]
    # sectors the device has something to do for: the DEM is larger than the zero frame and there
    # is a usable ray pixel (else zeros, or NaN inside the frame, as in _sx_rolling)
    todo = []
    for k, (window, _, _, dist) in enumerate(sectors):
        if ny <= 2 * window or nx <= 2 * window:
            continue
        if dist.size == 0 or np.all(np.isnan(dist)):
            outs[k][window : ny - window, window : nx - window] = np.nan
            continue
        todo.append(k)
    if todo:
        first, dj, di, dist, window = device.pack_sectors([sectors[k] for k in todo])
        planes = (C.c_void_p * len(todo))(*[_lib.ptr(outs[k]) for k in todo])
        _lib.check(_lib.lib().topo_amd_sx_multi_f32(
            _lib.ptr(src), ny, nx, len(todo), first.ctypes.data_as(_lib._i32p), dj.ctypes.data_as(_lib._i32p),
            di.ctypes.data_as(_lib._i32p), dist.ctypes.data_as(_lib._f64p), window.ctypes.data_as(_lib._i32p),
            float(height), planes), "topo_amd_sx_multi_f32")
    return [o.astype(dem.dtype, copy=False) for o in outs]


# ---- valley / ridge index ---------------------------------------------------------------------
def _valley_kernels(size, flat_list):
    """One normalised V / U profile per flat fraction (reference topo.py:456-492): distance to the
    centre row, repeated along the columns; a fraction ``f`` flattens the band of half-width
    ``int(floor(floor(size * f / 2) + 0.5))`` to the value on its edge.  Like the reference, every
    plane is brought to mean 0 / std 1 again after each fraction is applied, so later planes are
    flattened on normalised values (this order decides the float32 result).  Odd sizes only."""
    size = int(size)
    middle = size // 2
    if 2 * middle + 1 != size:
        # the reference fails here too: its profile has 2 * (size // 2) + 1 rows and cannot be
        # broadcast to (size, size) (topo.py:477-482); scale_to_pixel only ever yields odd sizes
        raise ValueError(f"valley / ridge kernels need an odd size, got {size}: operands could not be broadcast "
                         f"together with remapped shapes ({2 * middle + 1},{size}) -> ({size},{size})")
    rows = np.abs(np.arange(-middle, middle + 1)).astype(np.float32)
    kernels = np.tile(rows[None, :, None], (len(flat_list), 1, size))
    for ind, flat in enumerate(flat_list):
        half = int(np.floor(np.floor(size * flat / 2) + 0.5))
        kernels[ind, middle - half:middle + half + 1, :] = kernels[ind, middle - half, 0]
        kernels = (kernels - kernels.mean(axis=(1, 2), keepdims=True)) / kernels.std(axis=(1, 2), keepdims=True)
    return kernels


def _ridge_kernels(size, flat_list):
    """reference topo.py:495-512"""
    return _valley_kernels(size, flat_list) * -1


def _rotate_kernels(kernel, angle):
    """The kernel stack turned by ``angle`` degrees in its (row, column) plane with scipy's
    quadratic-spline ``ndimage.rotate`` on an enlarged canvas; canvas cells outside the turned
    square (left at the fill value -9999) are excluded from the re-normalisation and end up 0
    (reference topo.py:515-525).  Host-side parameter preparation, like the Sx ray geometry."""
    from numpy import ma
    from scipy import ndimage

    turned = ndimage.rotate(kernel, angle, axes=(1, 2), reshape=True, order=2, mode="constant", cval=-9999)
    turned = ma.masked_array(turned, mask=turned == -9999)
    turned = (turned - turned.mean(axis=(1, 2), keepdims=True)) / turned.std(axis=(1, 2), keepdims=True)
    return ma.MaskedArray.filled(turned, 0).astype(np.float32)


def _valley_ridge_tables(kernels, angles):
    """Per angle, what the reference's 3-D ``signal.convolve(dem3, kernels_rot, mode="same")``
    (topo.py:436) applies to the DEM in each output plane, laid out for the C ABI.

    The DEM is broadcast to one identical plane per kernel plane, so along that axis the "same"
    convolution just adds the kernel planes that overlap: output plane ``i`` of ``L`` sees the sum of
    ``K[b]`` with ``0 <= i - b + (L - 1) // 2 < L`` (three planes: K0+K1, K0+K1+K2, K1+K2).  The
    sums are flipped in both axes (a convolution becomes a correlation) and interleaved as four
    floats per tap.  Returns (taps float32, ksize int32, angles float32)."""
    n = kernels.shape[0]
    if not 1 <= n <= 4:
        raise ValueError(f"valley_ridge: {n} flat fractions; 1 to 4 are supported")
    centre = (n - 1) // 2

    def one_angle(angle):
        turned = _rotate_kernels(kernels, angle).astype(np.float64)
        side = turned.shape[1]
        block = np.zeros((side, side, 4), dtype=np.float32)
        for i in range(n):
            total = sum(turned[b] for b in range(n) if 0 <= i - b + centre < n)
            block[:, :, i] = total[::-1, ::-1]
        return block.reshape(-1)

    # the angles are independent and scipy / numpy release the GIL in the rotation and the
    # re-normalisation: for large kernels (seconds per call) they are spread over host threads
    workers = min(32, os.cpu_count() or 1, len(angles))
    if kernels.shape[1] >= 65 and workers > 1:
        from concurrent.futures import ThreadPoolExecutor  # noqa: PLC0415
        with ThreadPoolExecutor(workers) as pool:
            taps = list(pool.map(one_angle, angles))
    else:
        taps = [one_angle(angle) for angle in angles]
    ksize = [int(round((t.size // 4) ** 0.5)) for t in taps]
    return (np.ascontiguousarray(np.concatenate(taps), dtype=np.float32), np.asarray(ksize, dtype=np.int32),
            np.ascontiguousarray(angles, dtype=np.float32))


def valley_ridge(dem, size, mode, flat_list=[0, 0.15, 0.3], sigma=None):  # noqa: B006 (the reference's default)
    """Valley or ridge index: for 180 directions, the response of the standardised DEM to V- and
    U-shaped kernels of side ``size`` turned to that direction; returns ``[norm, direction]``, the
    largest response clipped at 0 and the direction (degrees, 0 = W-E, 90 = S-N) it came from
    (reference topo.py:389-447).  The 180 x ``len(flat_list)`` convolutions run in one pass over
    the DEM on the GPU."""
    if mode not in ("valley", "ridge"):
        raise ValueError(f"Unknown mode {mode!r}")
    values, _ = _unwrap(dem)
    _check_2d(values, "valley_ridge")
    field = globals()["dem"](values, sigma) if sigma else values
    src = _lib.as_f32(field)
    # the reference standardises with numpy's own float32 mean / std of the whole array (topo.py:427)
    mean, stdev = float(field.mean()), float(field.std())
    kernels = _ridge_kernels(size, flat_list) if mode == "ridge" else _valley_kernels(size, flat_list)
    taps, ksize, angles = _valley_ridge_tables(kernels, np.arange(0, 180, dtype=np.float32))
    norm = np.empty(src.shape, dtype=np.float32)
    direction = np.empty(src.shape, dtype=np.float32)
    _lib.check(_lib.lib().topo_amd_valley_ridge_f32(
        src.ctypes.data_as(_lib._vp), src.shape[0], src.shape[1], taps.ctypes.data_as(_lib._vp),
        ksize.ctypes.data_as(_lib._i32p), angles.ctypes.data_as(_lib._vp), ksize.size, kernels.shape[0], mean, stdev,
        norm.ctypes.data_as(_lib._vp), direction.ctypes.data_as(_lib._vp)), "topo_amd_valley_ridge_f32")
    return [norm, direction]
