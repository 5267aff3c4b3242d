"""CPU oracle for the per-pixel descriptor hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this module.  The product (``topo_descriptors_amd``) never does: it fails
loudly when the HIP library is missing instead of falling back to anything in here.

What this is
------------
A numpy/scipy restatement of the algorithms of MeteoSwiss/topo-descriptors'
``topo_descriptors/topo.py`` and ``helpers.py`` for the path tpi / std / gradient /
sobel / dem(gaussian) / sx.  The reference delegates its arithmetic to third-party
wheels that are not vendored and not version-pinned by it (``requirements.txt:1-8``):

* ``scipy.signal.convolve``  (FFT branch)     - topo.py:175, :301-302
* ``scipy.ndimage.gaussian_filter``           - topo.py:80, :173, :298, :631-635
* ``scipy.ndimage.convolve``                  - topo.py:682-683
* ``numpy.gradient`` and ufuncs               - topo.py:631-642
* ``numba.njit`` loop                         - topo.py:928-953

Each public function has two evaluators:

``*_scipy``  issues the same third-party calls in the same order and dtypes as the
             reference, so it reproduces the reference's float32-FFT noise.  It is what
             the CPU baseline times (single threaded, "1 core").
``*_exact``  evaluates the same mathematical formula directly in float64 (no FFT, no
             intermediate float32 rounding).  It is the noise-free target the tolerance
             contract of SURVEY.md section 8 needs next to the reference output.

Pinning
-------
Parity is pinned: ``tests/golden/*.npz`` were produced by importing the real reference
from ``/root/reference`` (numpy 2.2.6 / scipy 1.15.3) with ``tests/golden/make_golden.py``
and ``tests/test_oracle_golden.py`` checks every function here against them, plus the
four known-answer tests the reference itself carries (test/test_topo.py:6-67,
test/test_helpers.py:6-11).
"""

from __future__ import annotations

import numpy as np
from scipy import ndimage, signal

SCALE_STD = 4  # topo_descriptors/config/topo_descriptors.conf:5
MIN_ELEVATION = -100  # topo_descriptors/config/topo_descriptors.conf:2


# ----------------------------------------------------------------------------------------
# helpers.py restatements
# ----------------------------------------------------------------------------------------
def round_up_to_odd(values):
    """Nearest odd integer, numpy half-to-even rounding (helpers.py:108-111)."""
    v = np.asarray(values, dtype=np.float64)
    half_steps = np.round((v - 1.0) * 0.5)
    return np.asarray(2.0 * half_steps + 1.0, dtype=np.int64)


def grid_resolution(x_coords, y_coords):
    """Per-node grid spacing in coordinate units, signed (helpers.py:98-100).

    1-D coordinates give 1-D spacings (projected grids); 2-D coordinate meshes give 2-D
    spacings (the WGS84->UTM branch, helpers.py:91-97, whose reprojection is done by the
    absent ``utm`` wheel and is therefore taken as an input here).
    """
    x_coords = np.asarray(x_coords)
    y_coords = np.asarray(y_coords)
    x_res = np.gradient(x_coords, axis=x_coords.ndim - 1)
    y_res = np.gradient(y_coords, axis=0)
    return {"x": x_res, "y": y_res}


def scale_to_pixel(scales, x_coords, y_coords):
    """Metres -> odd pixel diameters, plus the resolution dict (helpers.py:68-105)."""
    res = grid_resolution(x_coords, y_coords)
    mean_res = np.mean(np.abs([res["x"].mean(), res["y"].mean()]))
    return round_up_to_odd(np.array(scales) / mean_res), res


def get_sigmas(smth_factors, scales_pxl):
    """Gaussian sigmas in pixels; falsy factor -> None (helpers.py:114-134)."""
    out = []
    for fact, px in zip(smth_factors, np.asarray(scales_pxl)):
        out.append(None if not fact else fact * px / SCALE_STD)
    return out


# ----------------------------------------------------------------------------------------
# disc kernel, TPI, STD
# ----------------------------------------------------------------------------------------
def circular_kernel(size):
    """0/1 float32 disc mask of diameter ``size`` (topo.py:191-213).

    ``m = int(size/2)``; tap (a, b) is set when (a-m)^2 + (b-m)^2 <= m^2; sizes below 5
    are all-ones squares.  Even sizes give an off-centre mask, kept on purpose.
    """
    size = int(size)
    m = size // 2
    if size < 5:
        return np.ones((size, size), dtype=np.float32)
    ax = np.arange(size) - m
    d2 = ax[:, None] ** 2 + ax[None, :] ** 2
    return (d2 <= m * m).astype(np.float32)


def disc_taps(size, drop_centre=False):
    """Tap offsets (dj, di) of the *correlation* form of the reference convolution.

    ``signal.convolve(a, k, "same")`` returns ``full[(K-1)//2 : (K-1)//2 + N]`` per axis,
    i.e. ``out[j] = sum_t k[t] * a[j + (K-1)//2 - t]`` with zero padding.  A tap at kernel
    index ``t`` therefore reads the DEM at offset ``(K-1)//2 - t``.  For odd sizes the mask
    is symmetric and this is the obvious centred disc; for even sizes it matters.
    """
    k = circular_kernel(size)
    if drop_centre:
        k[size // 2, size // 2] = 0.0
    c = (size - 1) // 2
    tj, ti = np.nonzero(k)
    return c - tj, c - ti, k


def tpi_scipy(dem, size, sigma=None):
    """TPI with the reference's own call sequence (topo.py:168-181)."""
    _, _, k = disc_taps(size, drop_centre=True)
    if sigma:
        dem = ndimage.gaussian_filter(dem, sigma)
    neighbourhood = signal.convolve(dem, k, mode="same")
    return dem - neighbourhood / np.sum(k)


def _disc_sum_f64(field64, size, drop_centre):
    """Zero-padded disc sum evaluated tap by tap in float64."""
    dj, di, _ = disc_taps(size, drop_centre)
    ny, nx = field64.shape
    pad = int(size)
    big = np.zeros((ny + 2 * pad, nx + 2 * pad), dtype=np.float64)
    big[pad : pad + ny, pad : pad + nx] = field64
    acc = np.zeros((ny, nx), dtype=np.float64)
    for a, b in zip(dj, di):
        acc += big[pad + a : pad + a + ny, pad + b : pad + b + nx]
    return acc, len(dj)


def tpi_exact(dem, size, sigma=None):
    """Same formula as :func:`tpi_scipy`, float64 direct evaluation."""
    field = np.asarray(dem, dtype=np.float64)
    if sigma:
        # the reference rounds the smoothed DEM to float32 before the disc sum
        field = ndimage.gaussian_filter(np.asarray(dem), sigma).astype(np.float64)
    total, n = _disc_sum_f64(field, size, drop_centre=True)
    if n == 0:
        with np.errstate(divide="ignore", invalid="ignore"):
            return field - total / 0.0
    return field - total / n


def std_scipy(dem, size, sigma=None):
    """Windowed sample standard deviation, reference call sequence (topo.py:295-307).

    Quirk kept: the squared field is ``trunc(dem)`` as int32 squared (topo.py:300) while
    the plain sum uses the untruncated float32 DEM.  Returns float64.
    """
    k = circular_kernel(size)
    n = np.sum(k)
    if sigma:
        dem = ndimage.gaussian_filter(dem, sigma)
    sq = dem.astype("int32") ** 2
    s1 = signal.convolve(dem, k, mode="same")
    s2 = signal.convolve(sq, k, mode="same")
    var = (s2 - s1**2 / n) / (n - 1)
    return np.sqrt(np.clip(var, 0, None))


def std_exact(dem, size, sigma=None):
    """The formula of :func:`std_scipy` in exact float64 (int64 for the squares)."""
    field = np.asarray(dem)
    if sigma:
        field = ndimage.gaussian_filter(field, sigma)
    trunc = np.trunc(field.astype(np.float64))
    s1, n = _disc_sum_f64(field.astype(np.float64), size, drop_centre=False)
    s2, _ = _disc_sum_f64(trunc * trunc, size, drop_centre=False)
    with np.errstate(divide="ignore", invalid="ignore"):
        var = (s2 - s1 * s1 / n) / (n - 1)
    return np.sqrt(np.clip(var, 0, None))


# ----------------------------------------------------------------------------------------
# Gaussian primitive, Sobel, gradient / slope / aspect
# ----------------------------------------------------------------------------------------
def gaussian_scipy(dem, sigma):
    """``ndimage.gaussian_filter`` as called at topo.py:80: reflect, truncate 4 sigma."""
    return ndimage.gaussian_filter(dem, sigma)


def gaussian_weights(sigma):
    """Normalised float64 taps scipy uses for one axis: radius int(4*sigma + 0.5)."""
    radius = int(4.0 * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1, dtype=np.float64)
    w = np.exp(-0.5 / (float(sigma) * float(sigma)) * x * x)
    return w / w.sum(), radius


def gaussian_exact(dem, sigma):
    """Separable Gaussian in float64 end to end (no float32 rounding between axes).

    ``sigma`` may be a scalar or an (axis0, axis1) pair like scipy's.
    """
    sig = np.broadcast_to(np.asarray(sigma, dtype=np.float64), (2,))
    out = np.asarray(dem, dtype=np.float64)
    for axis in (0, 1):
        if sig[axis] <= 1e-15:
            continue
        w, radius = gaussian_weights(sig[axis])
        padded = np.pad(out, [(radius, radius) if a == axis else (0, 0) for a in (0, 1)],
                        mode="symmetric")
        acc = np.zeros_like(out)
        n = out.shape[axis]
        for t in range(2 * radius + 1):
            sl = [slice(None), slice(None)]
            sl[axis] = slice(t, t + n)
            acc += w[t] * padded[tuple(sl)]
        out = acc
    return out


def sobel_scipy(dem):
    """3x3 Sobel pair normalised by 8, true convolution, reflect (topo.py:679-685)."""
    k = np.array([[1, 0, -1], [2, 0, -2], [1, 0, -1]], dtype=np.float32) / np.float32(8)
    return ndimage.convolve(dem, k), ndimage.convolve(dem, k.T)


def sobel_exact(dem):
    f = np.pad(np.asarray(dem, dtype=np.float64), 1, mode="symmetric")
    ny, nx = np.asarray(dem).shape
    def v(a, b):
        return f[1 + a : 1 + a + ny, 1 + b : 1 + b + nx]
    # convolution flips the kernel: dx = (right column - left column) weighted 1,2,1 over 8
    dx = (v(-1, 1) + 2 * v(0, 1) + v(1, 1) - v(-1, -1) - 2 * v(0, -1) - v(1, -1)) / 8.0
    dy = (v(1, -1) + 2 * v(1, 0) + v(1, 1) - v(-1, -1) - 2 * v(-1, 0) - v(-1, 1)) / 8.0
    return dx, dy


def _divide_by_resolution(dx, dy, res_meters):
    """In-place division by signed grid spacing (topo.py:707-712)."""
    y_res = np.asarray(res_meters["y"])
    if y_res.ndim == 1:
        y_res = y_res[:, None]
    dx /= np.asarray(res_meters["x"])
    dy /= y_res


def gradient_scipy(dem, sigma, res_meters, sig_ratio=1):
    """[dx, dy, slope, aspect] with the reference's branches (topo.py:628-644)."""
    if sigma <= 1:
        dx, dy = sobel_scipy(dem)
    elif sig_ratio == 1:
        dy, dx = np.gradient(ndimage.gaussian_filter(dem, sigma))
    else:
        perp = sigma * sig_ratio
        dx = np.gradient(ndimage.gaussian_filter(dem, (perp, sigma)), axis=1)
        dy = np.gradient(ndimage.gaussian_filter(dem, (sigma, perp)), axis=0)
    _divide_by_resolution(dx, dy, res_meters)
    slope = np.arctan(np.sqrt(dx**2 + dy**2)) * (180 / np.pi)
    aspect = (180 + np.degrees(np.arctan2(dx, dy))) % 360
    return [dx, dy, slope, aspect]


def gradient_exact(dem, sigma, res_meters, sig_ratio=1):
    """Float64 evaluation of the same branches."""
    if sigma <= 1:
        dx, dy = sobel_exact(dem)
    elif sig_ratio == 1:
        dy, dx = np.gradient(gaussian_exact(dem, sigma))
    else:
        perp = sigma * sig_ratio
        dx = np.gradient(gaussian_exact(dem, (perp, sigma)), axis=1)
        dy = np.gradient(gaussian_exact(dem, (sigma, perp)), axis=0)
    dx = np.array(dx, dtype=np.float64)
    dy = np.array(dy, dtype=np.float64)
    _divide_by_resolution(dx, dy, res_meters)
    slope = np.degrees(np.arctan(np.hypot(dx, dy)))
    aspect = (180.0 + np.degrees(np.arctan2(dx, dy))) % 360.0
    return [dx, dy, slope, aspect]


def wrapped_angle_diff(a, b):
    """min(d, 360-d) for angles in degrees."""
    d = np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) % 360.0
    return np.minimum(d, 360.0 - d)


# ----------------------------------------------------------------------------------------
# Sx
# ----------------------------------------------------------------------------------------
# Valley / ridge index (topo.py:389-531)
# ----------------------------------------------------------------------------------------
def valley_kernels(size, flat_list):
    """Normalised V / U profiles, one plane per flat fraction (topo.py:456-492).

    A plane is |row - middle| repeated along the columns (a valley running along x); a flat
    fraction f levels the band of half-width ``int(floor(floor(size*f/2) + 0.5))`` around the
    centre row to the value on its edge.  The reference re-normalises *all* planes (mean 0,
    std 1 over each plane) inside the loop over the fractions, after levelling each one
    (topo.py:484-490), so later planes are levelled on already normalised values; the same
    order of operations is kept here because it decides the float32 bits."""
    size = int(size)
    middle = size // 2
    if 2 * middle + 1 != size:  # the reference's broadcast_to fails the same way (topo.py:477-482)
        raise ValueError(f"operands could not be broadcast together: ({2 * middle + 1},{size}) -> ({size},{size})")
    profile = np.abs(np.arange(-middle, middle + 1)).astype(np.float32)
    plane = np.repeat(profile[:, None], size, axis=1)
    kernels = np.repeat(plane[None, :, :], len(flat_list), axis=0).copy()
    for ind, flat in enumerate(flat_list):
        half = int(np.floor(np.floor(size * flat / 2) + 0.5))
        kernels[ind, middle - half:middle + half + 1, :] = kernels[ind, middle - half, 0]
        kernels = (kernels - np.mean(kernels, axis=(1, 2), keepdims=True)) / np.std(
            kernels, axis=(1, 2), keepdims=True)
    return kernels


def ridge_kernels(size, flat_list):
    """topo.py:495-512."""
    return valley_kernels(size, flat_list) * -1


def rotate_kernels(kernels, angle):
    """topo.py:515-525: quadratic-spline rotation in the (row, column) plane with the output
    enlarged to hold the rotated square, cells outside it marked with -9999, statistics taken
    over the marked-in cells only, marked-out cells set to 0, float32."""
    import numpy.ma as ma

    rot = ndimage.rotate(kernels, angle, axes=(1, 2), reshape=True, order=2, mode="constant", cval=-9999)
    rot = ma.masked_array(rot, mask=rot == -9999)
    rot = (rot - np.mean(rot, axis=(1, 2), keepdims=True)) / np.std(rot, axis=(1, 2), keepdims=True)
    return ma.MaskedArray.filled(rot, 0).astype(np.float32)


def valley_ridge_plane_sums(kernels_rot):
    """What the reference's 3-D ``signal.convolve(dem3, kernels_rot, mode="same")`` applies to
    the DEM in each output plane (topo.py:436): the DEM is broadcast to ``L`` identical planes,
    so along the plane axis the "same" convolution adds up the kernel planes that overlap,
    out[i] = dem (*) sum of K[b] for 0 <= i - b + (L-1)//2 < L.  For the default three planes
    that is K0+K1, K0+K1+K2, K1+K2.  Returned in float64."""
    n = kernels_rot.shape[0]
    c = (n - 1) // 2
    k64 = kernels_rot.astype(np.float64)
    return [sum(k64[b] for b in range(n) if 0 <= i - b + c < n) for i in range(n)]


def _valley_ridge_base(mode, size, flat_list):
    if mode not in ("valley", "ridge"):
        raise ValueError(f"Unknown mode {mode!r}")
    return ridge_kernels(size, flat_list) if mode == "ridge" else valley_kernels(size, flat_list)


def normalise_dem(dem, sigma=None, stats=None):
    """topo.py:424-427: optional Gaussian pre-smooth, then (dem - mean) / std of the whole array,
    in the array's own precision.  ``stats=(mean, std)`` supplies the statistics from outside (a row
    block of a sharded DEM is standardised with those of the whole DEM)."""
    field = ndimage.gaussian_filter(dem, sigma) if sigma else dem
    if stats is None:
        return (field - field.mean()) / field.std()
    return (field - field.dtype.type(stats[0])) / field.dtype.type(stats[1])


def valley_ridge_scipy(dem, size, mode, flat_list=(0, 0.15, 0.3), sigma=None):
    """The reference's call sequence (topo.py:420-447) with the same third-party calls."""
    base = _valley_ridge_base(mode, size, list(flat_list))
    field = normalise_dem(dem, sigma)
    ny, nx = field.shape
    stack = np.broadcast_to(field, (len(flat_list), ny, nx))
    norm = np.zeros((ny, nx), dtype=np.float32) - np.inf
    direction = np.empty((ny, nx), dtype=np.float32)
    for angle in np.arange(0, 180, dtype=np.float32):
        conv = np.max(signal.convolve(stack, rotate_kernels(base, angle), mode="same"), axis=0)
        better = conv > norm
        norm[better] = conv[better]
        direction[better] = angle
    return [np.ndarray.clip(norm, min=0), direction]


def valley_ridge_exact(dem, size, mode, flat_list=(0, 0.15, 0.3), sigma=None, angles=None,
                       return_maps=False, stats=None, method="fft"):
    """Float64 evaluation of the same index from the same float32 kernels and the same float32
    normalised DEM: per angle the maximum over the plane sums of a float64 2-D convolution.
    ``return_maps`` also returns the per-angle maxima (n_angles x ny x nx), which is what a
    direction has to be judged against (the arg-max over 180 near-equal candidates is not
    stable under rounding).  ``method="direct"`` sums the taps pixel by pixel instead of going
    through an FFT, so a pixel's value does not depend on the size of the array around it."""
    base = _valley_ridge_base(mode, size, list(flat_list))
    field = np.asarray(normalise_dem(dem, sigma, stats), dtype=np.float64)
    angles = np.arange(0, 180, dtype=np.float32) if angles is None else np.asarray(angles, dtype=np.float32)
    maps = np.empty((len(angles),) + field.shape, dtype=np.float64)
    for a, angle in enumerate(angles):
        sums = valley_ridge_plane_sums(rotate_kernels(base, angle))
        if method == "direct":
            maps[a] = np.max([signal.convolve(field, k, mode="same", method="direct") for k in sums], axis=0)
        else:
            maps[a] = np.max([signal.fftconvolve(field, k, mode="same") for k in sums], axis=0)
    best = np.argmax(maps, axis=0)  # first maximum, like the reference's strict ">" update
    norm = np.clip(np.take_along_axis(maps, best[None], axis=0)[0], 0, None)
    direction = angles[best].astype(np.float64)
    return ([norm, direction], maps) if return_maps else [norm, direction]


# ----------------------------------------------------------------------------------------
def sx_distance(radius, dx, dy):
    """Metric distance of every cell of the search window to its centre (topo.py:861-878)."""
    rad_px = max(radius / abs(dy), radius / abs(dx))
    span = 2 * rad_px + 1
    centre = np.floor(span / 2)
    cells = np.arange(span)
    cols, rows = np.meshgrid(cells, cells)
    return np.sqrt(((rows - centre) * dy) ** 2 + ((cols - centre) * dx) ** 2)


def sx_source_idx_delta(azimuths, radius, dx, dy):
    """Index offset of the far end of each ray (topo.py:881-892); signed resolutions."""
    az = np.deg2rad(np.asarray(azimuths, dtype=np.float64))
    rows = np.rint(radius / dy * np.cos(az))
    cols = np.rint(radius / dx * np.sin(az))
    return np.stack([rows, cols], axis=1).astype(np.int64)


def sx_bresenhamlines(start, end):
    """Pixels strictly between each start point and ``end`` (topo.py:895-925).

    Per ray: unit steps along the dominant axis, nearest-integer rounding (numpy
    half-to-even), steps 1..max_iter where max_iter is the longest ray, truncated where
    the L1 distance to ``end`` stops decreasing, and with ``end`` itself removed.  Rays
    are concatenated in input order; duplicates are kept.
    """
    start = np.asarray(start)
    end = np.asarray(end)
    delta = end - start
    longest = int(np.max(np.abs(delta)))
    out = []
    for s, d in zip(start, delta):
        major = np.max(np.abs(d))
        unit = d.astype(np.float64) / major if major != 0 else np.zeros(d.shape)
        prev_l1 = None
        for step in range(1, longest + 1):
            p = np.rint(s + unit * step).astype(start.dtype)
            l1 = int(np.abs(p - end).sum())
            # the reference keeps a point while the L1 distance is non-increasing
            # (diff with prepend => the first point is always kept)
            if prev_l1 is not None and l1 > prev_l1:
                prev_l1 = l1
                continue
            prev_l1 = l1
            if np.all(p == end):
                continue
            out.append(p)
    if not out:
        return np.zeros((0, start.shape[-1]), dtype=start.dtype)
    return np.array(out, dtype=start.dtype)


def sx_geometry(azimuth, radius, dx, dy, azimuth_arc=10.0, azimuth_steps=15, radius_min=0.0):
    """Host-side geometry of topo.py:828-853 -> (window, offsets (P,2), distances (P,)).

    ``window`` is the zero-frame width ``int(W/2)``; offsets are relative to the target
    pixel, in ray order with duplicates; distances are NaN below ``radius_min``.
    """
    if azimuth_arc == 0:
        azimuth_steps = 1
    azimuths = np.linspace(azimuth - azimuth_arc / 2, azimuth + azimuth_arc / 2, azimuth_steps)
    dist = sx_distance(radius, dx, dy)
    dist[dist < radius_min] = np.nan
    centre = np.floor(np.array(dist.shape) / 2)
    source = (centre + sx_source_idx_delta(azimuths, radius, dx, dy)).astype(int)
    lines = sx_bresenhamlines(source, centre)
    window = int(dist.shape[0] / 2)
    lines = lines.astype(np.int64)
    d = dist[lines[:, 0], lines[:, 1]] if len(lines) else np.zeros((0,))
    return window, lines - window, d


def sx_rolling(dem, window, offsets, distances, height):
    """Max elevation angle over the offset table (topo.py:928-953), vectorised.

    Float64 arithmetic like the numba-compiled reference; result in the DEM's dtype; a
    frame of ``window`` pixels stays zero; NaN distances are skipped (nanmax).
    """
    dem = np.asarray(dem)
    ny, nx = dem.shape
    out = np.zeros_like(dem)
    if ny <= 2 * window or nx <= 2 * window:
        return out
    core = dem[window : ny - window, window : nx - window].astype(np.float64) + height
    best = np.full(core.shape, -np.inf)
    seen_any = np.zeros(core.shape, dtype=bool)
    for (oj, oi), d in zip(offsets, distances):
        if np.isnan(d):
            continue
        view = dem[window + oj : ny - window + oj, window + oi : nx - window + oi]
        with np.errstate(divide="ignore", invalid="ignore"):
            ang = np.rad2deg(np.arctan((view.astype(np.float64) - core) / d))
        ok = ~np.isnan(ang)
        best = np.where(ok & (ang > best), ang, best)
        seen_any |= ok
    best[~seen_any] = np.nan
    out[window : ny - window, window : nx - window] = best.astype(dem.dtype)
    return out


def sx(dem, x_coords, y_coords, azimuth, radius, height=10.0, azimuth_arc=10.0,
       azimuth_steps=15, radius_min=0.0):
    """Full Sx of topo.py:776-858 on a bare array plus its 1-D grid coordinates."""
    res = grid_resolution(x_coords, y_coords)
    dx = res["x"].mean()
    dy = res["y"].mean()
    window, offs, dist = sx_geometry(azimuth, radius, dx, dy, azimuth_arc, azimuth_steps,
                                     radius_min)
    return sx_rolling(dem, window, offs, dist, height)


# ----------------------------------------------------------------------------------------
# synthetic terrain used by tests and the CPU-baseline sample
# ----------------------------------------------------------------------------------------
def synthetic_dem(ny, nx, seed=0, integer=True, row0=0, col0=0):
    """Seeded terrain-like field: low-frequency sinusoids + 5 m noise, ~900-2900 m.

    Integer-valued metres stored as float32 by default (SRTM-like).
    ``row0/col0`` offset the window inside a conceptually infinite field so that shards
    can be generated independently and still agree on the smooth part.
    """
    rng = np.random.default_rng(seed)
    jj = (np.arange(ny, dtype=np.float64) + row0)[:, None]
    ii = (np.arange(nx, dtype=np.float64) + col0)[None, :]
    z = (1900.0
         + 520.0 * np.sin(jj / 211.0) * np.cos(ii / 173.0)
         + 310.0 * np.sin((jj + 2.0 * ii) / 97.0)
         + 120.0 * np.cos((3.0 * jj - ii) / 41.0)
         + 40.0 * np.sin(jj / 9.0) * np.sin(ii / 7.0))
    z = z + rng.normal(0.0, 5.0, size=(ny, nx))
    if integer:
        z = np.rint(z)
    return np.ascontiguousarray(z, dtype=np.float32)
