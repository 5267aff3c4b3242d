"""Parity of the HIP path against the golden vectors of the real reference and against the
oracle's exact float64 evaluation.  Every call goes through the C ABI of libtopo_amd.so.

Tolerance contract (written out here):
  * TPI, dx, dy, slope, Sx:  max|gpu - ref| / max|ref| <= 1e-4 over the whole array; on top of
    that a tighter bound against the exact evaluation, because the zero-padded borders inflate
    max|ref| for TPI.
  * aspect: wrapped difference <= 1e-4 * 360 deg where slope > 0.1 deg; dx, dy compared everywhere.
  * STD: (i) |gpu - exact| <= 1e-4 * max|ref|; (ii) |gpu - ref| <= |ref - exact|_max + 1e-4 * max|ref|
    (the reference's own float32-FFT noise floor is stored with each fixture).
"""
import os

import numpy as np
import pytest

from oracle import topo_oracle as orc

pytestmark = pytest.mark.gpu

from topo_descriptors_amd import topo  # noqa: E402

REL = 1e-4


def rel_range(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.nanmax(np.abs(a - b)) / max(np.nanmax(np.abs(b)), 1e-30))


class FakeVar:
    def __init__(self, values, dims):
        self.values, self.dims = values, dims


class FakeDataset:
    def __init__(self, dem, x, y, crs="epsg:2056"):
        self._v = {"dem": FakeVar(dem, ("y", "x")), "x": FakeVar(x, ("x",)), "y": FakeVar(y, ("y",))}
        self.attrs = {"crs": crs}

    def __getitem__(self, k):
        return self._v[k]

    def __iter__(self):
        return iter(["dem"])


SIZES = (3, 5, 6, 7, 17, 65, 67)


@pytest.mark.parametrize("tag", ["int", "frac"])
@pytest.mark.parametrize("size", SIZES)
def test_tpi_vs_reference(golden, tag, size):
    g = golden("tpi_std")
    dem = g["dem_" + tag]
    ref = g[f"tpi_{tag}_s{size}"]
    got = topo.tpi(dem, size)
    assert got.dtype == np.float32 and got.shape == ref.shape
    assert rel_range(got, ref) <= REL
    exact = orc.tpi_exact(dem, size)
    # integer part summed exactly, fractional part in float32: float32 output rounding
    # (half an ulp of a ~1500 m border value = 6e-5 m) plus ~1e-4 m on fractional DEMs
    assert np.max(np.abs(got - exact)) <= 2.5e-4
    inner = (slice(size, -size), slice(size, -size))
    if exact[inner].size:
        assert np.max(np.abs(got[inner] - exact[inner])) <= REL * np.max(np.abs(exact[inner]))


@pytest.mark.parametrize("tag", ["int", "frac"])
@pytest.mark.parametrize("size", SIZES)
def test_std_vs_reference(golden, tag, size):
    g = golden("tpi_std")
    dem = g["dem_" + tag]
    ref = g[f"std_{tag}_s{size}"]
    floor = float(g[f"std_{tag}_s{size}_floor"])
    got = topo.std(dem, size)
    assert got.dtype == np.float64 and got.shape == ref.shape
    exact = orc.std_exact(dem, size)
    scale = np.max(np.abs(ref))
    assert np.max(np.abs(got - exact)) <= REL * scale            # (i)
    assert np.max(np.abs(got - ref)) <= floor + REL * scale        # (ii)


def test_tpi_std_fused_equals_separate(golden):
    g = golden("tpi_std")
    for tag in ("int", "frac"):
        dem = g["dem_" + tag]
        t, s = topo.tpi_std(dem, 17)
        # one exact pipeline behind all three entry points
        assert np.array_equal(t, topo.tpi(dem, 17))
        assert np.array_equal(s, topo.std(dem, 17))
        assert np.max(np.abs(t - orc.tpi_exact(dem, 17))) <= 2.5e-4


def test_tpi_std_with_presmoothing(golden):
    g = golden("tpi_std")
    for tag in ("int", "frac"):
        dem = g["dem_" + tag]
        assert rel_range(topo.tpi(dem, 7, sigma=1.75), g[f"tpi_{tag}_s7_sig1p75"]) <= REL
        ref = g[f"std_{tag}_s17_sig2p125"]
        floor = float(g[f"std_{tag}_s17_sig2p125_floor"])
        got = topo.std(dem, 17, sigma=2.125)
        assert np.max(np.abs(got - ref)) <= floor + REL * np.max(ref)


def test_tpi_size_one_non_finite():
    dem = orc.synthetic_dem(8, 9, seed=7)
    assert not np.any(np.isfinite(topo.tpi(dem, 1)))


def test_tpi_std_multi_scale_is_the_single_calls():
    """Several scales from one upload (topo_amd_tpi_std_multi_f32): every plane has the bits of the
    single call - ring kernel, marching kernel and general kernel sizes, with and without pre-smoothing."""
    dem = orc.synthetic_dem(300, 260, seed=17)
    dem[::7, ::5] += 0.25  # some fractional tiles
    sizes, sigmas = [5, 17, 33, 67, 6], [None, 1.5, None, None, None]
    tpis, stds = topo.tpi_std_multi(dem, sizes, sigmas)
    for k, (size, sigma) in enumerate(zip(sizes, sigmas)):
        t, s = topo.tpi_std(dem, size, sigma)
        assert np.array_equal(tpis[k], t, equal_nan=True), size
        assert np.array_equal(stds[k], s, equal_nan=True), size
    only_t, none = topo.tpi_std_multi(dem, [9, 67], want_std=False)
    assert none is None and np.array_equal(only_t[1], topo.tpi(dem, 67))
    none, only_s = topo.tpi_std_multi(dem, [9], want_tpi=False)
    assert none is None and np.array_equal(only_s[0], topo.std(dem, 9))


@pytest.mark.parametrize("key,src", [("gauss_int_0.75", "dem_int"), ("gauss_int_2.25", "dem_int"),
                                     ("gauss_int_3.25", "dem_int"), ("gauss_big_30.25", "dem_big"),
                                     ("gauss_small_8.0", "dem_small")])
def test_gaussian_vs_reference(golden, key, src):
    g = golden("gaussian")
    sigma = float(key.rsplit("_", 1)[1])
    ref = g[key]
    got = topo.dem(g[src], sigma)
    assert got.dtype == np.float32 and got.shape == ref.shape
    assert rel_range(got, ref) <= REL
    # much tighter in practice: a couple of float32 ulps of a ~2000 m field
    assert np.max(np.abs(got.astype(np.float64) - orc.gaussian_exact(g[src], sigma))) <= 1e-3


def test_gaussian_anisotropic_and_identity(golden):
    g = golden("gaussian")
    dem = g["dem_int"]
    got = topo.dem(dem, (3.25, 0.0))
    from scipy import ndimage
    assert np.max(np.abs(got - ndimage.gaussian_filter(dem, (3.25, 0.0)))) <= 1e-3
    got = topo.dem(dem, (0.0, 2.25))
    assert np.max(np.abs(got - ndimage.gaussian_filter(dem, (0.0, 2.25)))) <= 1e-3
    assert np.array_equal(topo.dem(dem, 0.0), dem)
    assert np.array_equal(topo.dem(dem, 0.1), ndimage.gaussian_filter(dem, 0.1))  # radius 0


@pytest.mark.parametrize("sigma", [0.75, 2.25, 3.25, 6.0, 12.0])
@pytest.mark.parametrize("nx", [300, 299, 301])  # (any width takes the matrix cores since round 3)
def test_gaussian_nan_footprint(sigma, nx):
    """A non-finite sample makes non-finite every output ndimage.gaussian_filter (topo.py:80) makes non-finite.  On
    the matrix cores (radius int(4 sigma + 0.5) from 4) EXACTLY those: the kernels evaluate 32
    outputs against a zero-padded band of taps (0 x NaN = NaN), mark the tiles that came out non-finite, and a repair
    pass takes them again over each output's own window (round 3; before, up to 37 more outputs along each axis).
    The vector-ALU kernels (radius below 4) pad their taps to a chunk of 8: up to 7 more outputs towards lower
    indices.  The accumulation offsets must not spread it further (a non-finite offset falls back to 0), and the
    finite outputs keep their accuracy."""
    from scipy import ndimage
    dem = orc.synthetic_dem(200, nx, seed=21)
    dem[100, 151] = np.nan
    dem[7, 290] = np.inf
    dem[150:153, 40] = -np.inf
    want = ndimage.gaussian_filter(dem, sigma)
    got = topo.dem(dem, sigma)
    bad_ref, bad = ~np.isfinite(want), ~np.isfinite(got)
    assert not np.any(bad_ref & ~bad)
    if int(4 * sigma + 0.5) >= 4:
        assert np.array_equal(bad, bad_ref), int(np.sum(bad & ~bad_ref))
        assert np.array_equal(np.isnan(got), np.isnan(want))  # and NaN where the reference has NaN, inf where it has inf
    else:
        allowed = ndimage.binary_dilation(bad_ref, structure=np.ones((17, 17), bool))
        assert not np.any(bad & ~allowed)
    ok = ~bad
    assert np.max(np.abs(got[ok] - want[ok])) <= 1e-3


def test_gaussian_nan_repair_is_partition_invariant():
    """Row blocks with the ghost rows the library asks for give the single block's bits around non-finite samples
    too (the repair pass works on the global tile grid like the kernels it follows)."""
    from topo_descriptors_amd import _lib, device as d, shard
    gny, nx, sigma = 420, 512, 3.25
    dem = orc.synthetic_dem(gny, nx, seed=77)
    dem[139, 200] = np.nan     # next to a block seam for 3 blocks (rows 140, 280)
    dem[300, 31:34] = np.nan   # across a 32-column tile boundary
    dev = d.DeviceArray.from_host(dem)
    whole = d.DeviceArray(gny, nx)
    d.Block(dev).gaussian(sigma, sigma, whole)
    d.sync()
    w = whole.to_host()
    up, down = shard.halo_rows(_lib.DESC_GAUSS, sigma)
    for nb in (2, 3):
        parts = []
        for row0, rows in shard.split_rows(gny, nb):
            lo, hi = max(0, row0 - up), min(gny, row0 + rows + down)
            part = d.DeviceArray.from_host(dem[lo:hi])
            out = d.DeviceArray(rows, nx)
            d.Block(part, row0=lo, gny=gny).gaussian(sigma, sigma, out, out_row0=row0, out_rows=rows)
            d.sync()
            parts.append(out.to_host())
            part.free()
            out.free()
        assert np.array_equal(np.concatenate(parts), w, equal_nan=True), nb
    from scipy import ndimage
    assert np.array_equal(np.isnan(w), np.isnan(ndimage.gaussian_filter(dem, sigma)))
    dev.free()
    whole.free()


def test_gaussian_fused_is_the_two_passes():
    """Radius 4 ... 47 with one sigma runs both passes in one kernel (gauss_fused_f16_kernel): the arithmetic of the
    two f16 kernels tile for tile, the intermediate in LDS.  Same bits as the two passes (TOPO_AMD_GAUSS_FUSED=0 in
    a child process), on shapes with partial bands, partial tiles, a width that is not a multiple of 64, a block
    narrower than one tile, and through row blocks."""
    import subprocess, sys, tempfile
    shapes = [(200, 512), (129, 68), (33, 20), (300, 1000), (64, 64)]
    sigmas = [1.0, 2.25, 3.25, 4.0, 5.0, 7.75, 8.0, 11.75]  # radii 4, 9, 13, 16 (64-column raw blocks), 20, 31, 32, 47 (32-column ones)
    with tempfile.TemporaryDirectory() as tmp:
        code = (
            "import sys, numpy as np\n"
            "sys.path.insert(0, %r)\n"
            "from topo_descriptors_amd import topo\n"
            "from oracle import topo_oracle as orc\n"
            "out = {}\n"
            "for k, (ny, nx) in enumerate(%r):\n"
            "    dem = orc.synthetic_dem(ny, nx, seed=40 + k)\n"
            "    for s in %r:\n"
            "        out['%%d_%%s' %% (k, s)] = topo.dem(dem, s)\n"
            "np.savez(%r, **out)\n"
        ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), shapes, sigmas, os.path.join(tmp, "two.npz"))
        env = dict(os.environ, TOPO_AMD_GAUSS_FUSED="0")
        subprocess.run([sys.executable, "-c", code], check=True, env=env)
        two = np.load(os.path.join(tmp, "two.npz"))
        for k, (ny, nx) in enumerate(shapes):
            dem = orc.synthetic_dem(ny, nx, seed=40 + k)
            for s in sigmas:
                assert np.array_equal(topo.dem(dem, s), two["%d_%s" % (k, s)]), (ny, nx, s)
    # row blocks of the fused route: the single block's bits (clean DEM, and one with a non-finite sample, where the
    # two-pass kernels and their repair passes take over inside the library)
    from topo_descriptors_amd import _lib, device as d, shard
    gny, nx = 420, 512
    for poison, sigma in ((False, 3.25), (True, 3.25), (False, 6.0), (True, 6.0), (False, 10.0), (True, 10.0)):
        dem = orc.synthetic_dem(gny, nx, seed=78)
        if poison:
            dem[139, 200] = np.nan
            dem[300, 31:34] = np.inf
        dev = d.DeviceArray.from_host(dem)
        whole = d.DeviceArray(gny, nx)
        d.Block(dev).gaussian(sigma, sigma, whole)
        d.sync()
        w = whole.to_host()
        up, down = shard.halo_rows(_lib.DESC_GAUSS, sigma)
        for nb in (2, 3):
            parts = []
            for row0, rows in shard.split_rows(gny, nb):
                lo, hi = max(0, row0 - up), min(gny, row0 + rows + down)
                part = d.DeviceArray.from_host(dem[lo:hi])
                out = d.DeviceArray(rows, nx)
                d.Block(part, row0=lo, gny=gny).gaussian(sigma, sigma, out, out_row0=row0, out_rows=rows)
                d.sync()
                parts.append(out.to_host())
                part.free()
                out.free()
            assert np.array_equal(np.concatenate(parts), w, equal_nan=True), (poison, nb)
        dev.free()
        whole.free()


@pytest.mark.parametrize("sigma", [4.0, 4.25, 5.25, 6.25, 9.0, 30.25])
def test_gaussian_matrix_core_group_tails(sigma):
    """The matrix-core kernels take 4 MFMA steps (8 samples) per group, 4 groups per loop pass and the last 0-3
    groups in a tail: radii 16, 17, 21, 25, 36 and 121 leave 0, 1, 2, 3, 1 and 3 groups there.  Width 512: a
    multiple of 4 (the kernels' condition), four column strips, one of them partial rows; 200 rows: tiles cut
    by the block edge."""
    dem = orc.synthetic_dem(200, 512, seed=31)
    got = topo.dem(dem, sigma)
    assert np.max(np.abs(got.astype(np.float64) - orc.gaussian_exact(dem, sigma))) <= 1e-3


@pytest.mark.parametrize("sigma", [12.5, 20.0, 30.25])  # radii 50, 80, 121: 5, 7 and 9 steps of 32 window columns
def test_gaussian_split_once_axis1(sigma):
    """Radius 49 ... 121: axis 1 runs the split-once kernel (gauss_axis1_s1_kernel: samples split into their f16 pair once,
    against the reference of their 64-column slab, 16-row bands, two waves per SIMD).  Against the float64 filter on
    shapes with a partial band (rows % 16), a partial tile (columns % 32), fewer columns than a window and few enough
    rows for the bands to be cut into several runs; non-finite and huge samples: scipy's mask exactly; and against the
    tile kernels (TOPO_AMD_GAUSS_SPLIT_ONCE=0 in a child process): the two routes differ by float32 rounding only."""
    import subprocess, sys, tempfile
    from scipy import ndimage
    shapes = [(203, 1530), (37, 100), (16, 4100), (129, 321)]
    dems = [orc.synthetic_dem(ny, nx, seed=60 + k) for k, (ny, nx) in enumerate(shapes)]
    got = [topo.dem(d, sigma) for d in dems]
    for d, g in zip(dems, got):
        assert np.max(np.abs(g.astype(np.float64) - orc.gaussian_exact(d, sigma))) <= 1e-3
        g1 = topo.dem(d, (0.0, sigma))
        assert np.max(np.abs(g1.astype(np.float64) - orc.gaussian_exact(d, (0.0, sigma)))) <= 6e-4
    bad = dems[0].copy()
    bad[100, 700] = np.nan
    bad[5, 64] = np.inf        # a slab's reference sample (column 64)
    bad[150, 128] = 3.0e9      # another one, finite but huge
    bad[60:63, 1529] = -np.inf
    want = ndimage.gaussian_filter(bad, sigma)
    res = topo.dem(bad, sigma)
    assert np.array_equal(np.isnan(res), np.isnan(want)) and np.array_equal(np.isfinite(res), np.isfinite(want))
    bad[150, 128] = dems[0][150, 128]  # (values: without the huge sample, whose float32 products swamp its neighbourhood)
    want = ndimage.gaussian_filter(bad, sigma)
    res = topo.dem(bad, sigma)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(res), fin) and np.max(np.abs(res[fin] - want[fin])) <= 2e-3
    with tempfile.TemporaryDirectory() as tmp:
        code = (
            "import sys, numpy as np\n"
            "sys.path.insert(0, %r)\n"
            "from topo_descriptors_amd import topo\n"
            "from oracle import topo_oracle as orc\n"
            "np.savez(%r, *[topo.dem(orc.synthetic_dem(ny, nx, seed=60 + k), %r) for k, (ny, nx) in enumerate(%r)])\n"
        ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(tmp, "tile.npz"), sigma, shapes)
        subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, TOPO_AMD_GAUSS_SPLIT_ONCE="0"))
        tile = np.load(os.path.join(tmp, "tile.npz"))
        for k, g in enumerate(got):
            t = tile["arr_%d" % k]
            assert np.max(np.abs(g - t)) <= 1e-3 and not np.array_equal(g, t)  # (another kernel did run)


def test_gaussian_turned_tile_order_against_the_float64_filter():
    """The axis-1 matrix-core kernels (fused, tile, split-once) let every band / row block walk its row from a tile of its own
    (no lockstep over the columns: profiles/r04_pitch_spread.txt); a run that meets the end of the row goes on at its start.
    Round 4 proved the bits equal to the in-step order, whose switch is retired; here the turned order against the float64
    filter on DEMs wide enough for the turn to be on (64 tiles of 32 columns / 32 tiles of 64), with a partial last tile."""
    from scipy import ndimage
    for k, (ny, nx) in enumerate([(40, 4100), (70, 2500)]):
        dem = orc.synthetic_dem(ny, nx, seed=70 + k)
        for sigma in (3.25, 10.0, 13.0, 30.25):
            want = ndimage.gaussian_filter(dem.astype(np.float64), sigma, mode="reflect")
            got = topo.dem(dem, sigma)
            assert np.max(np.abs(got - want)) <= 6e-4, (nx, sigma, float(np.max(np.abs(got - want))))


def test_gaussian_of_a_raster_beyond_the_f16_range():
    """the f16 matrix-core kernels stage samples beyond +-1e5 as 0 and leave their outputs to the repair pass;
    a raster whose ordinary values lie out there (a DEM in millimetres) would be repaired pixel by pixel.  The library
    samples a block at its first Gaussian / gradient call and gives such a raster to the vector-ALU kernels: the same
    bits as with the matrix-core routes switched off, and float32 accuracy at that magnitude.  A NaN sea does not count."""
    import subprocess, sys, tempfile
    dem = orc.synthetic_dem(300, 520, seed=88) * np.float32(1000.0)  # 1.5e6 ... 3e6
    sigmas = [3.25, 13.0]
    got = [topo.dem(dem, s) for s in sigmas]
    grad = topo.gradient(dem, 3.25, {"x": 25.0, "y": -25.0})
    for s, g in zip(sigmas, got):
        assert np.max(np.abs(g.astype(np.float64) - orc.gaussian_exact(dem, s))) <= 2.0  # (0.25 is one float32 ulp out there)
    with tempfile.TemporaryDirectory() as tmp:
        code = (
            "import sys, numpy as np\n"
            "sys.path.insert(0, %r)\n"
            "from topo_descriptors_amd import topo\n"
            "from oracle import topo_oracle as orc\n"
            "dem = orc.synthetic_dem(300, 520, seed=88) * np.float32(1000.0)\n"
            "np.savez(%r, *([topo.dem(dem, s) for s in %r] + list(topo.gradient(dem, 3.25, {'x': 25.0, 'y': -25.0}))))\n"
        ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(tmp, "valu.npz"), sigmas)
        env = dict(os.environ, TOPO_AMD_GAUSS_MFMA_MIN_RADIUS="1000", TOPO_AMD_GRAD_MFMA_MIN_RADIUS="1000")
        subprocess.run([sys.executable, "-c", code], check=True, env=env)
        valu = np.load(os.path.join(tmp, "valu.npz"))
        for k, g in enumerate(got + list(grad)):
            assert np.array_equal(g, valu["arr_%d" % k], equal_nan=True), k
    sea = orc.synthetic_dem(300, 520, seed=88)
    sea[:, :200] = np.nan  # 38 % of the samples: still the matrix-core route with its repair pass
    from scipy import ndimage
    res, want = topo.dem(sea, 13.0), ndimage.gaussian_filter(sea, 13.0)
    assert np.array_equal(np.isnan(res), np.isnan(want))
    fin = np.isfinite(want)
    assert np.max(np.abs(res[fin] - want[fin])) <= 2e-3


def check_gradient(got, ref_by_name, exact=None):
    dx, dy, slope, aspect = got
    for a in got:
        assert a.dtype == np.float32
    assert rel_range(dx, ref_by_name["dx"]) <= REL
    assert rel_range(dy, ref_by_name["dy"]) <= REL
    assert rel_range(slope, ref_by_name["slope"]) <= REL
    steep = ref_by_name["slope"] > 0.1
    if np.any(steep):
        d = orc.wrapped_angle_diff(aspect, ref_by_name["aspect"])
        assert np.max(d[steep]) <= REL * 360.0
    assert np.all((aspect >= 0) & (aspect < 360))


GRAD_CASES = [("sob_n", 0.75, "n", 1), ("g3_n", 3.25, "n", 1), ("g3_s", 3.25, "s", 1),
              ("g3_2d", 3.25, "2d", 1), ("g3_r2_n", 3.25, "n", 2), ("g2_r05_n", 2.25, "n", 0.5)]


@pytest.mark.parametrize("tag,sigma,res_tag,ratio", GRAD_CASES)
def test_gradient_vs_reference(golden, tag, sigma, res_tag, ratio):
    g = golden("gradient")
    res = {"x": g[f"res_{res_tag}_x"], "y": g[f"res_{res_tag}_y"]}
    dem = g["dem_int"].copy()
    got = topo.gradient(dem, sigma, res, sig_ratio=ratio)
    assert np.array_equal(dem, g["dem_int"])  # input not mutated
    check_gradient(got, {n: g[f"{tag}_{n}"] for n in ("dx", "dy", "slope", "aspect")})


def test_gradient_large_sigma(golden):
    """sigma = 30.25 on a 300 x 280 DEM: the smoothed field is so smooth that the float32
    rounding of it (done by the reference after each axis, topo.py:631) quantises dy in steps
    of 2 ulp / 60 m = 8e-6, i.e. 0.6e-4 of max|dy| here.  One differing rounding already costs
    more than 1e-4 range-normalised, so this case uses the two-sided form of the contract:
    |gpu - ref| <= |ref - exact|_max + 1e-4 max|ref|, and gpu vs exact on its own."""
    g = golden("gradient")
    res = {"x": g["res_b_x"], "y": g["res_b_y"]}
    got = topo.gradient(g["dem_big"], 30.25, res)
    exact = orc.gradient_exact(g["dem_big"], 30.25, res)
    for k, nm in enumerate(("dx", "dy", "slope")):
        ref = g[f"g30_big_{nm}"]
        floor = float(g[f"g30_big_{nm}_floor"])
        scale = np.max(np.abs(ref))
        assert np.max(np.abs(got[k] - ref)) <= floor + REL * scale, nm
        assert np.max(np.abs(got[k] - exact[k])) <= floor + REL * scale, nm
    steep = g["g30_big_slope"] > 0.1
    d = orc.wrapped_angle_diff(got[3], g["g30_big_aspect"])
    assert np.max(d[steep]) <= float(g["g30_big_aspect_floor"]) + REL * 360.0


def test_sobel_vs_reference(golden):
    g = golden("gradient")
    dx, dy = topo.sobel(g["dem_int"])
    assert rel_range(dx, g["sobel_dx"]) <= 1e-6 and rel_range(dy, g["sobel_dy"]) <= 1e-6


def test_aspect_conventions(golden):
    g = golden("gradient")
    res = {"x": g["res_f_x"], "y": g["res_f_y"]}
    flat = topo.gradient(g["plane_flat_in"], 2.0, res)
    assert np.all(flat[2] == 0) and np.all(flat[3] == 0)      # -0.0 dy keeps aspect at 0
    north = topo.gradient(g["plane_northf_in"], 2.0, res)
    assert np.max(orc.wrapped_angle_diff(north[3], g["plane_northf_aspect"])) <= 1e-3
    east = topo.gradient(g["plane_eastf_in"], 2.0, res)
    assert np.max(np.abs(east[3] - g["plane_eastf_aspect"])) <= 1e-3
    for got, tag in ((north, "northf"), (east, "eastf")):
        assert rel_range(got[2], g[f"plane_{tag}_slope"]) <= REL


SX_TAGS = ["az0", "az90", "az225", "arc0", "rmin", "south_up", "aniso"]


@pytest.mark.parametrize("tag", SX_TAGS)
def test_sx_vs_reference(golden, tag):
    g = golden("sx")
    az, radius, height, arc, steps, rmin = g[f"{tag}_params"]
    ds = FakeDataset(g["dem"], g[f"{tag}_x"], g[f"{tag}_y"])
    got = topo.sx(ds, az, radius, height=height, azimuth_arc=arc, azimuth_steps=int(steps),
                  radius_min=rmin)
    ref = g[f"{tag}_out"]
    assert got.dtype == np.float32 and got.shape == ref.shape
    assert rel_range(got, ref) <= REL
    assert np.array_equal(got == 0, ref == 0)  # the zero frame


def test_sx_nan_handling():
    dem = orc.synthetic_dem(64, 72, seed=11)
    dem[30, 40] = np.nan
    x = 2600000.0 + 30.0 * np.arange(72)
    y = 1200000.0 - 30.0 * np.arange(64)
    got = topo.sx(FakeDataset(dem, x, y), 0, 300.0)
    want = orc.sx(dem, x, y, 0, 300.0)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.max(np.abs(got[ok] - want[ok])) <= REL * np.max(np.abs(want[ok]))


def test_ragged_shapes_and_tiny_inputs():
    # shapes that are not multiples of any tile, down to a single row / column
    for ny, nx in ((1, 1), (1, 37), (53, 1), (33, 129), (97, 257), (130, 64)):
        dem = orc.synthetic_dem(ny, nx, seed=ny * 1000 + nx)
        for size in (3, 7, 17):
            got = topo.tpi(dem, size)
            assert np.max(np.abs(got - orc.tpi_exact(dem, size))) <= 2.5e-4, (ny, nx, size)
            s = topo.std(dem, size)
            e = orc.std_exact(dem, size)
            assert np.max(np.abs(s - e)) <= REL * max(np.max(e), 1.0), (ny, nx, size)
        sm = topo.dem(dem, 2.25)
        assert np.max(np.abs(sm - orc.gaussian_exact(dem, 2.25))) <= 1e-3, (ny, nx)


def test_long_filters_beyond_the_lds_tile():
    """sigma = 60 (radius 240): axis 1 runs as the wave-shift kernel, which has no LDS tile."""
    dem = orc.synthetic_dem(700, 1100, seed=31)
    got = topo.dem(dem, 60.0)
    assert np.max(np.abs(got - orc.gaussian_exact(dem, 60.0))) <= 1e-3
    x = 2600000.0 + 30.0 * np.arange(1100)
    y = 1200000.0 - 30.0 * np.arange(700)
    res = orc.grid_resolution(x, y)
    g = topo.gradient(dem, 60.0, res)
    e = orc.gradient_exact(dem, 60.0, res)
    for k in range(3):
        assert np.max(np.abs(g[k] - e[k])) <= 1e-4 * np.max(np.abs(e[k])) + 2e-5


def test_filters_wider_than_a_wavefront_can_chain():
    """sigma = 120 (radius 480): axis 1 goes through transpose + axis-0 kernel + transpose."""
    dem = orc.synthetic_dem(260, 300, seed=33)
    got = topo.dem(dem, 120.0)
    assert np.max(np.abs(got - orc.gaussian_exact(dem, 120.0))) <= 1e-3
    x = 2600000.0 + 30.0 * np.arange(300)
    y = 1200000.0 - 30.0 * np.arange(260)
    res = orc.grid_resolution(x, y)
    g = topo.gradient(dem, 120.0, res)
    e = orc.gradient_exact(dem, 120.0, res)
    # the float32 rounding of the smoothed field (a few 1e-4 m, the reference has the same) is
    # 2e-5 in dx, dy on a 30 m grid and 57.3 times that in the slope in degrees
    for k, floor in ((0, 2e-5), (1, 2e-5), (2, 57.3 * 2e-5)):
        assert np.max(np.abs(g[k] - e[k])) <= 1e-4 * np.max(np.abs(e[k])) + floor


def test_sx_window_beyond_the_lds_tile():
    """radius 6000 m on a 30 m grid: window 201 pixels, scanned straight from global memory."""
    dem = orc.synthetic_dem(460, 470, seed=35)
    x = 2600000.0 + 30.0 * np.arange(470)
    y = 1200000.0 - 30.0 * np.arange(460)
    got = topo.sx(FakeDataset(dem, x, y), 45.0, 6000.0)
    want = orc.sx(dem, x, y, 45.0, 6000.0)
    assert np.array_equal(got == 0, want == 0)
    assert np.max(np.abs(got - want)) <= REL * np.max(np.abs(want))


def test_page_locked_host_arrays_and_gate_statistic():
    """Round 4 additions to the C ABI: topo_amd_host_alloc hands out page-locked memory the host-buffer entry points take
    like any other array (same bits as with a pageable one), and topo_amd_gate_giveups answers (a count that only
    careful-mode shard calls of this process can have raised)."""
    import ctypes as C

    from topo_descriptors_amd import _lib

    lib = _lib.lib()
    ny, nx, size = 300, 512, 17
    dem = orc.synthetic_dem(ny, nx, seed=12)
    want = topo.tpi(dem, size)
    hin, hout = C.c_void_p(), C.c_void_p()
    _lib.check(lib.topo_amd_host_alloc(C.byref(hin), dem.nbytes), "host_alloc")
    _lib.check(lib.topo_amd_host_alloc(C.byref(hout), dem.nbytes), "host_alloc")
    try:
        pin_in = np.frombuffer((C.c_char * dem.nbytes).from_address(hin.value), dtype=np.float32).reshape(ny, nx)
        pin_out = np.frombuffer((C.c_char * dem.nbytes).from_address(hout.value), dtype=np.float32).reshape(ny, nx)
        pin_in[:] = dem
        _lib.check(lib.topo_amd_tpi_f32(hin, ny, nx, size, 0.0, hout), "topo_amd_tpi_f32")
        assert np.array_equal(pin_out, want)
        del pin_in, pin_out
    finally:
        _lib.check(lib.topo_amd_host_free(hin), "host_free")
        _lib.check(lib.topo_amd_host_free(hout), "host_free")
    n = C.c_uint(0xFFFFFFFF)
    _lib.check(lib.topo_amd_gate_giveups(C.byref(n)), "gate_giveups")
    assert n.value != 0xFFFFFFFF
