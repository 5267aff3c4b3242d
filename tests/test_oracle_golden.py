"""The CPU oracle against the golden vectors captured from the real reference.

Everything here runs without a GPU.  ``*_scipy`` evaluators issue the same third-party
calls as the reference, so they must reproduce the golden arrays bit for bit (same numpy /
scipy build) - a loose tolerance is still used so a different scipy build on the GPU box
does not turn FFT round-off into a failure.  ``*_exact`` evaluators must sit within the
recorded noise floor of the reference.
"""
import numpy as np
import pytest

from oracle import topo_oracle as orc


def rel_range(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.nanmax(np.abs(a - b)) / max(np.nanmax(np.abs(b)), 1e-30)


# ---- the reference's own four known-answer tests, restated literally -------------------
def test_ref_kat_sx_distance():  # reference test/test_topo.py:6-28
    out = orc.sx_distance(150.0, 50.0, 40.0)
    first = np.array([256.1249695, 219.31712199, 188.67962264, 167.63054614, 160.0,
                      167.63054614, 188.67962264, 219.31712199, 256.1249695])
    assert np.all(np.isclose(out[0, :], first))
    assert out.dtype == np.float64


def test_ref_kat_sx_bresenhamlines():  # reference test/test_topo.py:31-54
    out = orc.sx_bresenhamlines(np.array([[8, 9], [17, 22]]), np.array([15, 15]))
    expected = np.array([[9, 10], [10, 11], [11, 12], [12, 12], [13, 13], [14, 14],
                         [17, 21], [16, 20], [16, 19], [16, 18], [16, 17], [15, 16]])
    assert np.all(out == expected)
    assert out.dtype == np.int64


def test_ref_kat_sx_source_idx_delta():  # reference test/test_topo.py:57-67
    out = orc.sx_source_idx_delta(np.array([3.0, 4.0, 5.0, 6.0]), 500, 20, 30)
    assert np.all(out == np.array([[17, 1], [17, 2], [17, 2], [17, 3]]))
    assert out.dtype == np.int64


def test_ref_kat_round_up_to_odd():  # reference test/test_helpers.py:6-11
    out = orc.round_up_to_odd(np.arange(0.1, 10, 0.7))
    assert out.dtype == np.int64
    assert list(out) == [1, 1, 1, 3, 3, 3, 5, 5, 5, 7, 7, 7, 9, 9, 9]


# ---- helpers ----------------------------------------------------------------------------
def test_helpers_golden(golden):
    g = golden("helpers")
    assert np.array_equal(orc.round_up_to_odd(g["odd_in"]), g["odd_out"])
    for tag in ("30", "25"):
        px, res = orc.scale_to_pixel([2000, 200, 500], g["x" + tag], g["y" + tag])
        assert np.array_equal(px, g["px" + tag]) and px.dtype == np.int64
        assert np.array_equal(res["x"], g[f"res{tag}_x"])
        assert np.array_equal(res["y"], g[f"res{tag}_y"])
    sig = orc.get_sigmas([None, 0.5, 1, 0], np.array([67, 7, 17, 9]))
    want = g["sigmas"]
    for s, w in zip(sig, want):
        assert (s is None and np.isnan(w)) or s == w


def test_circular_kernel_golden(golden):
    g = golden("circular_kernel")
    counts = {}
    for key, ref in g.items():
        size = int(key[1:])
        mine = orc.circular_kernel(size)
        assert mine.dtype == np.float32 and np.array_equal(mine, ref)
        counts[size] = int(mine.sum())
    # tap counts quoted in SURVEY.md section 8a
    assert counts[3] == 9 and counts[5] == 13 and counts[7] == 29
    assert counts[17] == 197 and counts[65] == 3209 and counts[67] == 3409


# ---- TPI / STD --------------------------------------------------------------------------
SIZES = (3, 5, 6, 7, 17, 65, 67)


@pytest.mark.parametrize("tag", ["int", "frac"])
@pytest.mark.parametrize("size", SIZES)
def test_tpi_golden(golden, tag, size):
    g = golden("tpi_std")
    dem = g["dem_" + tag]
    ref = g[f"tpi_{tag}_s{size}"]
    got = orc.tpi_scipy(dem, size)
    assert got.dtype == ref.dtype == np.float32
    assert rel_range(got, ref) <= 1e-6
    exact = orc.tpi_exact(dem, size)
    floor = float(g[f"tpi_{tag}_s{size}_floor"])
    assert np.max(np.abs(exact - ref)) <= floor * 1.001 + 1e-12
    # the reference's float32 FFT noise is small for TPI: pin it
    assert floor / np.max(np.abs(ref)) < 1e-4


@pytest.mark.parametrize("tag", ["int", "frac"])
@pytest.mark.parametrize("size", SIZES)
def test_std_golden(golden, tag, size):
    g = golden("tpi_std")
    dem = g["dem_" + tag]
    ref = g[f"std_{tag}_s{size}"]
    got = orc.std_scipy(dem, size)
    assert got.dtype == ref.dtype == np.float64
    # sqrt amplifies FFT round-off near zero variance; compare variances on the scipy twin
    assert np.max(np.abs(got**2 - ref**2)) <= 1e-6 * max(np.max(ref**2), 1.0)
    exact = orc.std_exact(dem, size)
    floor = float(g[f"std_{tag}_s{size}_floor"])
    assert np.max(np.abs(exact - ref)) <= floor * 1.001 + 1e-12


def test_tpi_std_with_sigma_golden(golden):
    g = golden("tpi_std")
    for tag in ("int", "frac"):
        dem = g["dem_" + tag]
        assert rel_range(orc.tpi_scipy(dem, 7, sigma=1.75), g[f"tpi_{tag}_s7_sig1p75"]) <= 1e-6
        ref = g[f"std_{tag}_s17_sig2p125"]
        got = orc.std_scipy(dem, 17, sigma=2.125)
        assert np.max(np.abs(got**2 - ref**2)) <= 1e-6 * np.max(ref**2)


def test_tpi_size_one_is_non_finite():
    # size=1: the only tap is the zeroed centre -> division by zero
    dem = orc.synthetic_dem(8, 9, seed=7)
    with np.errstate(all="ignore"):
        out = orc.tpi_scipy(dem, 1)
    assert not np.any(np.isfinite(out))


# ---- Gaussian ---------------------------------------------------------------------------
def test_gaussian_golden(golden):
    g = golden("gaussian")
    for key, src in (("gauss_int_0.75", "dem_int"), ("gauss_int_2.25", "dem_int"),
                     ("gauss_int_3.25", "dem_int"), ("gauss_big_30.25", "dem_big"),
                     ("gauss_small_8.0", "dem_small")):
        sigma = float(key.rsplit("_", 1)[1])
        ref = g[key]
        got = orc.gaussian_scipy(g[src], sigma)
        assert got.dtype == np.float32 and np.array_equal(got, ref)
        exact = orc.gaussian_exact(g[src], sigma)
        floor = float(g[key + "_floor"])
        assert np.max(np.abs(exact - ref)) <= floor * 1.001 + 1e-12
        # float32 rounding of a ~2000 m field: about one ulp (1.2e-4..2.4e-4 m)
        assert floor < 5e-4


def test_gaussian_weights_match_scipy():
    from scipy.ndimage import gaussian_filter1d
    for sigma in (0.75, 3.25, 30.25):
        w, radius = orc.gaussian_weights(sigma)
        impulse = np.zeros(2 * radius + 1)
        impulse[radius] = 1.0
        assert np.allclose(gaussian_filter1d(impulse, sigma, mode="constant"), w, atol=1e-16)
    assert orc.gaussian_weights(3.25)[1] == 13 and orc.gaussian_weights(30.25)[1] == 121


# ---- gradient ---------------------------------------------------------------------------
GRAD_CASES = [("sob_n", 0.75, "n", 1), ("g3_n", 3.25, "n", 1), ("g3_s", 3.25, "s", 1),
              ("g3_2d", 3.25, "2d", 1), ("g3_r2_n", 3.25, "n", 2), ("g2_r05_n", 2.25, "n", 0.5)]


@pytest.mark.parametrize("tag,sigma,res_tag,ratio", GRAD_CASES)
def test_gradient_golden(golden, tag, sigma, res_tag, ratio):
    g = golden("gradient")
    res = {"x": g[f"res_{res_tag}_x"], "y": g[f"res_{res_tag}_y"]}
    got = orc.gradient_scipy(g["dem_int"], sigma, res, sig_ratio=ratio)
    exact = orc.gradient_exact(g["dem_int"], sigma, res, sig_ratio=ratio)
    for nm, a, e in zip(("dx", "dy", "slope", "aspect"), got, exact):
        ref = g[f"{tag}_{nm}"]
        assert a.dtype == np.float32
        assert np.array_equal(a, ref), nm
        floor = float(g[f"{tag}_{nm}_floor"])
        if nm == "aspect":
            assert np.max(orc.wrapped_angle_diff(e, ref)) <= floor * 1.001 + 1e-9
        else:
            assert np.max(np.abs(e - ref)) <= floor * 1.001 + 1e-12


def test_gradient_big_sigma_golden(golden):
    g = golden("gradient")
    res = {"x": g["res_b_x"], "y": g["res_b_y"]}
    got = orc.gradient_scipy(g["dem_big"], 30.25, res)
    for nm, a in zip(("dx", "dy", "slope", "aspect"), got):
        assert np.array_equal(a, g[f"g30_big_{nm}"])


def test_sobel_golden(golden):
    g = golden("gradient")
    dx, dy = orc.sobel_scipy(g["dem_int"])
    assert np.array_equal(dx, g["sobel_dx"]) and np.array_equal(dy, g["sobel_dy"])
    ex, ey = orc.sobel_exact(g["dem_int"])
    assert np.max(np.abs(ex - dx)) < 1e-3 and np.max(np.abs(ey - dy)) < 1e-3


def test_aspect_conventions(golden):
    g = golden("gradient")
    res = {"x": g["res_f_x"], "y": g["res_f_y"]}
    for tag in ("flat", "northf", "eastf"):
        got = orc.gradient_scipy(g[f"plane_{tag}_in"], 2.0, res)
        for nm, a in zip(("dx", "dy", "slope", "aspect"), got):
            assert np.array_equal(a, g[f"plane_{tag}_{nm}"])
    assert np.all(g["plane_flat_slope"] == 0) and np.all(g["plane_flat_aspect"] == 0)
    assert np.allclose(g["plane_northf_aspect"], 0.0) or np.allclose(g["plane_northf_aspect"] % 360, 0)
    assert np.allclose(g["plane_eastf_aspect"], 90.0)


# ---- Sx ---------------------------------------------------------------------------------
def test_sx_geometry_golden(golden):
    g = golden("sx_geometry")
    n_geo = sum(1 for k in g if k.startswith("dist") and k.endswith("_args"))
    n_az = sum(1 for k in g if k.startswith("az"))
    for m in range(n_geo):
        radius, dx, dy = g[f"dist{m}_args"]
        dist = orc.sx_distance(radius, dx, dy)
        assert dist.dtype == np.float64 and np.array_equal(dist, g[f"dist{m}"])
        centre = np.floor(np.array(dist.shape) / 2)
        for n in range(n_az):
            delta = orc.sx_source_idx_delta(g[f"az{n}"], radius, dx, dy)
            assert delta.dtype == np.int64 and np.array_equal(delta, g[f"delta_a{n}_g{m}"])
            lines = orc.sx_bresenhamlines((centre + delta).astype(int), centre)
            ref = g[f"lines_a{n}_g{m}"]
            assert lines.shape == ref.shape and np.array_equal(lines, ref), (n, m)
    assert np.array_equal(orc.sx_bresenhamlines(g["bres_start"], g["bres_end"]), g["bres_out"])


def test_sx_point_counts():
    # 240 points / 32 unique at r=500 m on a 30 m grid, window 35
    window, offs, dist = orc.sx_geometry(0.0, 500.0, 30.0, -30.0)
    assert window == 17 and len(offs) == 240
    assert len(np.unique(offs, axis=0)) == 32
    assert np.all(offs[:, 0] < 0) and np.all(np.abs(offs[:, 1]) <= 1)


SX_TAGS = ["az0", "az90", "az225", "arc0", "rmin", "south_up", "aniso"]


@pytest.mark.parametrize("tag", SX_TAGS)
def test_sx_golden(golden, tag):
    g = golden("sx")
    az, radius, height, arc, steps, rmin = g[f"{tag}_params"]
    got = orc.sx(g["dem"], g[f"{tag}_x"], g[f"{tag}_y"], az, radius, height=height,
                 azimuth_arc=arc, azimuth_steps=int(steps), radius_min=rmin)
    ref = g[f"{tag}_out"]
    assert got.dtype == ref.dtype == np.float32
    assert got.shape == ref.shape
    # the un-jitted reference loop evaluates z in float32 (NumPy 2 promotion), numba and
    # this oracle in float64: 1e-6 degrees apart at most
    assert np.max(np.abs(got.astype(np.float64) - ref)) <= 2e-5
    assert np.array_equal(got == 0, ref == 0)


# ---- valley / ridge index -------------------------------------------------------
VR_TAGS = ["int_valley_s7", "int_ridge_s7", "int_valley_s5", "int_valley_s17", "int_valley_s9_flat0",
           "int_ridge_s9_flat2", "frac_valley_s7", "frac_valley_s9_sig"]


def _vr_case(g, tag):
    p = g[f"{tag}_params"]
    size, mode, sigma, flats = int(p[0]), ("valley", "ridge")[int(p[1])], (None if p[2] < 0 else float(p[2])), list(p[3:])
    dem = g["dem_int"] if tag.startswith("int") else g["dem_frac"]
    return dem, size, mode, flats, sigma


def test_valley_kernels_golden(golden):
    """The kernels the reference builds, before and after rotation, bit for bit."""
    g = golden("valley_ridge")
    for size, flats in ((5, [0, 0.15, 0.3]), (7, [0, 0.15, 0.3]), (9, [0.2, 0.4]), (17, [0, 0.15, 0.3])):
        base = orc.valley_kernels(size, flats)
        want = g[f"kernels_s{size}_n{len(flats)}"]
        assert base.dtype == want.dtype and np.array_equal(base, want), (size, flats)
        assert np.array_equal(orc.ridge_kernels(size, flats), -want)
        for angle in (0, 1, 33, 45, 90, 137, 179):
            rot = orc.rotate_kernels(base, np.float32(angle))
            w = g[f"kernels_s{size}_n{len(flats)}_rot{angle}"]
            assert rot.shape == w.shape and rot.dtype == w.dtype, (size, angle)
            assert np.max(np.abs(rot - w)) <= 1e-6, (size, angle)  # spline arithmetic of another scipy build


def test_valley_ridge_plane_sums_are_what_the_3d_convolution_does():
    """The reference convolves a 3-plane broadcast of the DEM with the 3-plane kernel stack in 3-D
    (topo.py:436); per output plane that is a 2-D convolution with a sum of kernel planes."""
    from scipy import signal
    rng = np.random.default_rng(0)
    dem = rng.normal(size=(30, 36))
    for n in (1, 2, 3, 4):
        k = rng.normal(size=(n, 6, 7)).astype(np.float32)
        full = signal.convolve(np.broadcast_to(dem, (n, 30, 36)), k.astype(np.float64), mode="same", method="direct")
        for i, ksum in enumerate(orc.valley_ridge_plane_sums(k)):
            assert np.max(np.abs(full[i] - signal.convolve(dem, ksum, mode="same", method="direct"))) <= 1e-10, (n, i)


@pytest.mark.parametrize("tag", VR_TAGS)
def test_valley_ridge_golden(golden, tag):
    g = golden("valley_ridge")
    dem, size, mode, flats, sigma = _vr_case(g, tag)
    norm_ref, dir_ref = g[f"{tag}_norm"], g[f"{tag}_dir"]
    got = orc.valley_ridge_scipy(dem, size, mode, flats, sigma)
    assert got[0].dtype == np.float32 and got[1].dtype == np.float32
    assert rel_range(got[0], norm_ref) <= 1e-5
    assert np.mean(got[1] == dir_ref) >= 0.999
    # the float64 evaluation sits within the recorded floor, and every direction the reference
    # chose is a maximiser of the exact per-angle maps up to that floor
    (norm_ex, _), maps = orc.valley_ridge_exact(dem, size, mode, flats, sigma, return_maps=True)
    floor = float(g[f"{tag}_norm_floor"])
    assert np.max(np.abs(norm_ex - norm_ref)) <= floor * 1.001 + 1e-12
    at_ref_dir = np.take_along_axis(maps, dir_ref.astype(int)[None], axis=0)[0]
    assert np.max(np.max(maps, axis=0) - at_ref_dir) <= 4 * floor + 1e-6


def test_valley_ridge_unknown_mode():
    with pytest.raises(ValueError):
        orc.valley_ridge_scipy(np.zeros((8, 8), np.float32), 5, "canyon")


def test_valley_kernels_even_size_fails_like_the_reference():
    # reference topo.py:477-482: a (size + 1, size) profile cannot be broadcast to (size, size)
    for size in (4, 6, 8):
        with pytest.raises(ValueError):
            orc.valley_kernels(size, [0, 0.15, 0.3])
