"""Host-side logic of the product package and the shape of the C ABI.  No GPU needed:
nothing here launches a kernel."""
import ctypes
import os
import re

import numpy as np
import pytest

from topo_descriptors_amd import _lib, helpers as hlp, topo

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeVar:
    def __init__(self, values, dims):
        self.values = values
        self.dims = dims


class FakeDataset:
    """Duck-typed stand-in for xarray.Dataset (xarray is not installed in this image)."""

    def __init__(self, dem, x, y, crs="epsg:2056", dims=("y", "x")):
        self._v = {"dem": FakeVar(dem, dims), "x": FakeVar(x, ("x",)), "y": FakeVar(y, ("y",))}
        self.attrs = {} if crs is None else {"crs": crs}

    def __getitem__(self, k):
        return self._v[k]

    def __iter__(self):
        return iter(["dem"])


# ---- the reference's four known-answer tests against the PRODUCT helpers ------------------------
def test_sx_distance():  # reference test/test_topo.py:6-28
    out = topo._sx_distance(150.0, 50.0, 40.0)
    first = np.array([256.1249695, 219.31712199, 188.67962264, 167.63054614, 160.0,
                      167.63054614, 188.67962264, 219.31712199, 256.1249695])
    assert np.all(np.isclose(out[0, :], first))
    assert out.dtype == np.float64


def test_sx_bresenhamlines():  # reference test/test_topo.py:31-54
    out = topo._sx_bresenhamlines(np.array([[8, 9], [17, 22]]), np.array([15, 15]))
    expected = np.array([[9, 10], [10, 11], [11, 12], [12, 12], [13, 13], [14, 14],
                         [17, 21], [16, 20], [16, 19], [16, 18], [16, 17], [15, 16]])
    assert np.all(out == expected)
    assert out.dtype == np.int64


def test_sx_source_idx_delta():  # reference test/test_topo.py:57-67
    out = topo._sx_source_idx_delta(np.array([3.0, 4.0, 5.0, 6.0]), 500, 20, 30)
    assert np.all(out == np.array([[17, 1], [17, 2], [17, 2], [17, 3]]))
    assert out.dtype == np.int64


def test_round_up_to_odd():  # reference test/test_helpers.py:6-11
    outputs = hlp.round_up_to_odd(np.arange(0.1, 10, 0.7))
    assert outputs.dtype == np.int64
    assert list(outputs) == [1, 1, 1, 3, 3, 3, 5, 5, 5, 7, 7, 7, 9, 9, 9]


# ---- golden vectors ---------------------------------------------------------------------------
def test_geometry_helpers_golden(golden):
    g = golden("sx_geometry")
    n_geo = sum(1 for k in g if k.startswith("dist") and k.endswith("_args"))
    n_az = sum(1 for k in g if k.startswith("az"))
    for m in range(n_geo):
        radius, dx, dy = g[f"dist{m}_args"]
        dist = topo._sx_distance(radius, dx, dy)
        assert dist.shape == g[f"dist{m}"].shape
        assert np.allclose(dist, g[f"dist{m}"], rtol=1e-15, atol=1e-12)
        centre = np.floor(np.array(dist.shape) / 2)
        for n in range(n_az):
            delta = topo._sx_source_idx_delta(g[f"az{n}"], radius, dx, dy)
            assert np.array_equal(delta, g[f"delta_a{n}_g{m}"])
            lines = topo._sx_bresenhamlines((centre + delta).astype(int), centre)
            assert np.array_equal(lines, g[f"lines_a{n}_g{m}"])


def test_circular_kernel_golden(golden):
    g = golden("circular_kernel")
    for key, ref in g.items():
        got = topo.circular_kernel(int(key[1:]))
        assert got.dtype == np.float32 and np.array_equal(got, ref), key
    lib = _lib.load()
    for size, taps in ((3, 9), (5, 13), (7, 29), (17, 197), (65, 3209), (67, 3409), (81, 5025)):
        assert lib.topo_amd_disc_tap_count(size) == taps


def test_scale_to_pixel_and_sigmas_golden(golden):
    g = golden("helpers")
    for tag in ("30", "25"):
        ds = FakeDataset(np.zeros((40, 50), np.float32), g["x" + tag], g["y" + tag])
        px, res = hlp.scale_to_pixel([2000, 200, 500], ds)
        assert np.array_equal(px, g["px" + tag]) and px.dtype == np.int64
        assert np.array_equal(res["x"], g[f"res{tag}_x"]) and np.array_equal(res["y"], g[f"res{tag}_y"])
    sig = hlp.get_sigmas([None, 0.5, 1, 0], np.array([67, 7, 17, 9]))
    for s, w in zip(sig, g["sigmas"]):
        assert (s is None and np.isnan(w)) or s == w


# ---- WGS84 grids: the built-in UTM projection (reference helpers.py:91-97 calls the `utm` package, absent here) ---------------
# Known answers of the `utm` package itself: its README's example and the city table of its own test-suite (metres, rounded there).
UTM_README = ((51.2, 7.5), (395201.3103811303, 5673135.241182375, 32))
UTM_CITIES = [((50.77535, 6.08389), (294409, 5628898, 32)),       # Aachen
              ((40.71435, -74.00597), (583960, 4507523, 18)),      # New York
              ((-41.28646, 174.77624), (313784, 5427057, 60)),     # Wellington
              ((-33.92487, 18.42406), (261878, 6243186, 34)),      # Capetown
              ((-32.89018, -68.84405), (514586, 6360877, 19)),     # Mendoza
              ((64.83778, -147.71639), (466013, 7190568, 6)),      # Fairbanks
              ((56.79680, -5.00601), (377486, 6296562, 30))]       # Ben Nevis


def _kruger_utm(lat, lon, zone):
    """Independent check: Krueger's n-series (Karney 2011, to n^4), not the Snyder series the package and the product use."""
    a, f = 6378137.0, 1 / 298.257223563
    n = f / (2 - f)
    A = a / (1 + n) * (1 + n ** 2 / 4 + n ** 4 / 64)
    al = [n / 2 - 2 * n ** 2 / 3 + 5 * n ** 3 / 16 + 41 * n ** 4 / 180, 13 * n ** 2 / 48 - 3 * n ** 3 / 5 + 557 * n ** 4 / 1440,
          61 * n ** 3 / 240 - 103 * n ** 4 / 140, 49561 * n ** 4 / 161280]
    phi, lam = np.radians(lat), np.radians(lon - ((zone - 1) * 6 - 180 + 3))
    e = np.sqrt(f * (2 - f))
    t = np.sinh(np.arctanh(np.sin(phi)) - e * np.arctanh(e * np.sin(phi)))
    xi, eta = np.arctan2(t, np.cos(lam)), np.arctanh(np.sin(lam) / np.sqrt(1 + t * t))
    east = 500000 + 0.9996 * A * (eta + sum(al[j] * np.cos(2 * (j + 1) * xi) * np.sinh(2 * (j + 1) * eta) for j in range(4)))
    north = 0.9996 * A * (xi + sum(al[j] * np.sin(2 * (j + 1) * xi) * np.cosh(2 * (j + 1) * eta) for j in range(4)))
    return east, north + (10000000 if np.max(lat) < 0 else 0)


def test_utm_known_answers_of_the_package():
    (lat, lon), (e, n, z) = UTM_README
    east, north, zone = hlp._utm_from_latlon(np.array([lat]), np.array([lon]))
    assert zone == z and abs(east[0] - e) < 1e-6 and abs(north[0] - n) < 1e-6
    for (lat, lon), (e, n, z) in UTM_CITIES:
        east, north, zone = hlp._utm_from_latlon(np.array([lat]), np.array([lon]))
        assert zone == z and abs(east[0] - e) <= 0.5 and abs(north[0] - n) <= 0.5, (lat, lon)
    # the zone exceptions the package lists: Norway (32 V) and Svalbard (31, 33, 35, 37 X)
    for (lat, lon), z in [((60.0, 4.0), 32), ((56.0, 3.0), 32), ((55.999, 4.0), 31), ((64.0, 4.0), 31), ((72.0, 8.99), 31),
                          ((72.0, 9.0), 33), ((78.0, 20.99), 33), ((78.0, 21.0), 35), ((78.0, 33.0), 37), ((78.0, 42.0), 38),
                          ((0.0, -180.0), 1), ((0.0, 179.99), 60), ((46.8, 8.2), 32)]:
        assert hlp._utm_zone_number(np.array([lat]), np.array([lon])) == z, (lat, lon)


def test_utm_against_the_kruger_series_and_one_zone_for_an_array():
    rng = np.random.default_rng(11)
    for south in (False, True):
        lat = rng.uniform(0.5, 83.5, 4000) * (-0.95 if south else 1)
        zone0 = None
        lon = rng.uniform(6.0, 12.0, 4000)                        # inside zone 32 (the first element decides)
        lat[0], lon[0] = (-10.0 if south else 46.0), 9.0
        east, north, zone = hlp._utm_from_latlon(lat, lon)
        assert zone == 32
        ke, kn = _kruger_utm(lat, lon, 32)
        assert np.max(np.abs(east - ke)) < 5e-3 and np.max(np.abs(north - kn)) < 5e-3   # the package's series: millimetres in-zone
    # a grid across a zone boundary stays in the FIRST element's zone (eastings keep growing: np.gradient of them is the resolution)
    lon, lat = np.meshgrid(np.linspace(5.0, 7.0, 41), np.linspace(47.0, 46.0, 21))
    east, north, zone = hlp._utm_from_latlon(lat, lon)
    assert zone == 31 and np.all(np.diff(east, axis=1) > 0) and np.all(np.diff(north, axis=0) < 0)
    ke, kn = _kruger_utm(lat, lon, 31)
    assert np.max(np.abs(east - ke)) < 0.05 and np.max(np.abs(north - kn)) < 0.05
    # the package's range errors
    for lat, lon in [(84.1, 0.0), (-80.1, 0.0), (0.0, 180.1), (0.0, -180.1)]:
        with pytest.raises(ValueError):
            hlp._utm_from_latlon(np.array([lat]), np.array([lon]))
    with pytest.raises(ValueError):
        hlp._utm_from_latlon(np.array([-1.0, 1.0]), np.array([7.0, 7.0]))   # "latitudes must all have the same sign"


def test_scale_to_pixel_of_a_wgs84_grid():
    """reference helpers.py:89-104: meshgrid, from_latlon, float32, np.gradient along the last axis / axis 0."""
    x = 7.0 + np.arange(60) / 1200.0                               # 3 arc seconds
    y = 47.0 - np.arange(45) / 1200.0
    ds = FakeDataset(np.zeros((45, 60), np.float32), x, y, crs="EPSG:4326")
    px, res = hlp.scale_to_pixel([2000, 200, 500], ds)
    assert res["x"].shape == (45, 60) and res["y"].shape == (45, 60) and res["x"].dtype == np.float32
    lon, lat = np.meshgrid(x, y)
    ke, kn = _kruger_utm(lat, lon, 32)
    want_x, want_y = np.gradient(ke.astype(np.float32), axis=1), np.gradient(kn.astype(np.float32), axis=0)
    assert np.max(np.abs(res["x"] - want_x)) <= 0.07 and np.max(np.abs(res["y"] - want_y)) <= 0.6   # float32 metres: ulp 1/32 and 1/2
    assert np.all(res["x"] > 0) and np.all(res["y"] < 0)
    mean_res = np.mean(np.abs([res["x"].mean(), res["y"].mean()]))
    assert abs(mean_res - (63.3 + 92.65) / 2) < 0.3               # 3" at 47 N: ~63.3 m east-west, ~92.65 m north-south
    assert np.array_equal(px, hlp.round_up_to_odd(np.array([2000, 200, 500]) / mean_res)) and list(px) == [25, 3, 7]


# ---- error behaviour of the boundary ------------------------------------------------------------
def test_check_dem_errors():
    x, y = np.arange(5.0), np.arange(4.0)
    dem = np.zeros((4, 5), np.float32)
    with pytest.raises(ValueError):
        hlp.check_dem(dem)                                  # not a Dataset
    with pytest.raises(ValueError):
        hlp.check_dem(FakeDataset(dem, x, y, dims=("x", "y")))
    with pytest.raises(KeyError):
        hlp.check_dem(FakeDataset(dem, x, y, crs=None))
    with pytest.raises(ValueError):
        hlp.check_dem(FakeDataset(dem, x, y, crs="swiss grid"))
    hlp.check_dem(FakeDataset(dem, x, y, crs="EPSG:2056"))
    assert hlp.get_da(FakeDataset(dem, x, y)).values is dem


def test_sx_requires_dataset():
    with pytest.raises(TypeError):
        topo.sx(np.zeros((8, 8), np.float32), 0, 500.0)


def test_halo_rows_contract():
    lib = _lib.load()
    up, down = ctypes.c_int32(), ctypes.c_int32()
    # Gaussian radii 4 ... 15 run on the matrix cores, whose 32-row tiles take their accumulation offset 16 rows
    # into the tile: 16 ghost rows (gradient: 17) instead of R (R + 1)
    cases = [(_lib.DESC_TPI, 67, 0, 33, 33), (_lib.DESC_TPI, 6, 0, 3, 2), (_lib.DESC_STD, 7, 1.75, 3 + 16, 3 + 16),
             (_lib.DESC_STD, 7, 0.75, 3 + 3, 3 + 3), (_lib.DESC_STD, 7, 5.0, 3 + 20, 3 + 20),
             (_lib.DESC_GAUSS, 30.25, 0, 121, 121), (_lib.DESC_GAUSS, 3.25, 0, 16, 16), (_lib.DESC_GAUSS, 0.75, 0, 3, 3),
             (_lib.DESC_GRADIENT, 30.25, 0, 122, 122), (_lib.DESC_GRADIENT, 3.25, 0, 17, 17),
             (_lib.DESC_GRADIENT, 1.5, 0, 17, 17), (_lib.DESC_GRADIENT, 3.25, 1.0, 17, 17),
             (_lib.DESC_GRADIENT, 3.25, 2.0, 27, 27), (_lib.DESC_GRADIENT, 2.25, 0.5, 10, 10),
             (_lib.DESC_GRADIENT, 0.75, 0, 1, 1), (_lib.DESC_SOBEL, 0, 0, 1, 1),
             (_lib.DESC_SX, 17, 0, 17, 0)]
    for desc, p0, p1, a, b in cases:
        assert lib.topo_amd_halo_rows(desc, p0, p1, ctypes.byref(up), ctypes.byref(down)) == 0
        assert (up.value, down.value) == (a, b), (desc, p0, p1)


# ---- the C ABI: library loads and exports every declared symbol ----------------------------
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "topo_amd.h")).read()
    declared = set(re.findall(r"\b(topo_amd_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 40
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/topo_amd.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert b"gfx950" in _lib.load().topo_amd_version()


def test_no_cpu_fallback_without_gpu():
    lib = _lib.load()
    if lib.topo_amd_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.TopoAmdError):
        topo.tpi(np.zeros((8, 8), np.float32), 3)
    # entry points refuse to run before topo_amd_init
    assert lib.topo_amd_sync() != 0
    assert b"topo_amd_init" in lib.topo_amd_last_error()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "topo_descriptors_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(root, f)).read()
                assert "oracle" not in text.lower() or f == "__init__.py" and False, os.path.join(root, f)


def test_valley_kernels_match_the_reference_kernels_and_errors(golden):
    """Host-side kernel construction of topo.valley_ridge (no GPU needed): bit for bit the
    reference's kernels (golden fixture), the table layout of the C ABI, ValueError for even sizes
    (the reference's broadcast fails, topo.py:477-482) and for an unknown mode (topo.py:421-422)."""
    g = golden("valley_ridge")
    for size, flats in ((5, [0, 0.15, 0.3]), (7, [0, 0.15, 0.3]), (9, [0.2, 0.4]), (17, [0, 0.15, 0.3])):
        base = topo._valley_kernels(size, flats)
        want = g[f"kernels_s{size}_n{len(flats)}"]
        assert base.dtype == want.dtype and np.array_equal(base, want)
        assert np.array_equal(topo._ridge_kernels(size, flats), -want)
        for angle in (0, 1, 33, 45, 90, 137, 179):
            rot = topo._rotate_kernels(base, np.float32(angle))
            w = g[f"kernels_s{size}_n{len(flats)}_rot{angle}"]
            assert rot.shape == w.shape and np.max(np.abs(rot - w)) <= 1e-6
    taps, ksize, angles = topo._valley_ridge_tables(topo._valley_kernels(7, [0, 0.15, 0.3]),
                                                    np.arange(0, 180, dtype=np.float32))
    assert taps.dtype == np.float32 and ksize.dtype == np.int32 and len(ksize) == 180 and len(angles) == 180
    assert taps.size == 4 * int((ksize.astype(np.int64) ** 2).sum())
    assert np.all(taps.reshape(-1, 4)[:, 3] == 0)  # three planes: the fourth component stays empty
    for size in (4, 6, 8):
        with pytest.raises(ValueError):
            topo._valley_kernels(size, [0, 0.15, 0.3])
    with pytest.raises(ValueError):
        topo.valley_ridge(np.zeros((8, 8), np.float32), 5, "canyon")


def test_the_helpers_either_side_of_the_path(caplog):
    from topo_descriptors_amd import helpers as hlp

    @hlp.timer
    def work(a, b=2):
        """doc"""
        return a * b

    with caplog.at_level("INFO", logger="topo_descriptors_amd.helpers"):
        assert work(3, b=4) == 12
    assert work.__name__ == "work" and work.__doc__ == "doc"
    assert any("Computed in" in r.getMessage() for r in caplog.records)

    nan = np.nan
    a = np.array([[nan, 1.0, nan, nan, 4.0, nan],
                  [nan, nan, nan, nan, nan, nan],
                  [7.0, nan, nan, nan, nan, nan],
                  [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]], dtype=np.float32)
    f = hlp.fill_na_array(a)
    assert f.dtype == np.float32 and np.isnan(a[0, 0])  # a copy
    assert f[0, 0] == 1.0 and f[0, 5] == 4.0             # edge values beyond the valid samples
    assert f[0, 2] == 1.0 and f[0, 3] == 4.0             # nearest valid sample along x
    assert np.isnan(f[1]).all() and np.isnan(f[2, 1:]).all() and f[2, 0] == 7.0  # too few samples: left alone
    assert np.array_equal(f[3], a[3])
    if hlp._xr is None:
        for call in (lambda: hlp.get_dem_netcdf("dem.nc"), lambda: hlp.fill_na(None),
                     lambda: hlp.to_netcdf(a, None, "x")):
            with pytest.raises(ImportError, match="xarray"):
                call()


def test_fill_na_array_known_answers():
    """helpers.fill_na_array: the NaN fill of the reference's fill_na (helpers.py:137-154:
    ``interpolate_na(dim="x", method="nearest", fill_value="extrapolate")``), answers derived by hand:
    interior gap (nearest valid sample along x, the left one on a tie), leading / trailing gap (edge value),
    all-NaN row (stays NaN), and no coupling between rows."""
    from topo_descriptors_amd import helpers as hlp

    nan = np.nan
    a = np.array([[1.0, nan, nan, 4.0, nan, 6.0],      # 1 -> x=0 (1 away, x=3 is 2 away); 2 -> x=3; 4: tie 3|5 -> left
                  [nan, nan, 3.0, 5.0, nan, nan],      # leading gap -> 3, trailing gap -> 5
                  [nan, nan, nan, nan, nan, nan],      # nothing to take from: stays NaN
                  [7.0, 8.0, 9.0, 10.0, 11.0, 12.0],   # untouched
                  [nan, 2.0, nan, nan, nan, 9.0]],     # 2 -> x=1 (1 away); 3: tie 1|5 -> left; 4 -> x=5
                 dtype=np.float32)
    want = np.array([[1.0, 1.0, 4.0, 4.0, 4.0, 6.0],
                     [3.0, 3.0, 3.0, 5.0, 5.0, 5.0],
                     [nan, nan, nan, nan, nan, nan],
                     [7.0, 8.0, 9.0, 10.0, 11.0, 12.0],
                     [2.0, 2.0, 2.0, 2.0, 9.0, 9.0]], dtype=np.float32)
    got = hlp.fill_na_array(a)
    assert got.dtype == a.dtype and got is not a
    assert np.array_equal(got, want, equal_nan=True)
    assert np.isnan(a[0, 1])  # the input is not modified
    # uneven coordinates decide "nearest" by distance in x, not by index
    x = np.array([0.0, 1.0, 2.0, 10.0, 11.0, 12.0])
    b = np.array([[5.0, nan, nan, nan, nan, 8.0]])
    assert np.array_equal(hlp.fill_na_array(b, x), [[5.0, 5.0, 5.0, 8.0, 8.0, 8.0]])
    # and the indices the reference returns next to the filled DEM are just np.where(isnan)
    ind = np.where(np.isnan(a))
    assert np.isnan(a[ind]).all() and len(ind[0]) == 3 + 4 + 6 + 4


def test_a_dataarray_like_is_not_a_dataset():
    """The reference's sx / check_dem insist on a Dataset (topo.py:825-826 TypeError, helpers.py:179-180
    ValueError); a DataArray quacks almost like one (attrs, __getitem__, __iter__) and must be refused."""
    from topo_descriptors_amd import helpers as hlp, topo

    class Var:
        def __init__(self, values, dims):
            self.values, self.dims = values, dims

    class DatasetLike:
        data_vars = {"dem": None}
        attrs = {"crs": "epsg:2056"}

        def __init__(self):
            self._v = {"dem": Var(np.zeros((4, 5), np.float32), ("y", "x")), "x": Var(np.arange(5.0), ("x",)),
                       "y": Var(-np.arange(4.0), ("y",))}

        def __getitem__(self, k):
            return self._v[k]

        def __iter__(self):
            return iter(["dem"])

    class DataArrayLike:
        dims = ("y", "x")
        attrs = {"crs": "epsg:2056"}
        values = np.zeros((4, 5), np.float32)

        def __getitem__(self, k):
            return self.values[k]

        def __iter__(self):
            return iter(self.values)

    hlp.check_dem(DatasetLike())
    assert not hlp._looks_like_dataset(DataArrayLike())
    with pytest.raises(ValueError, match="xr.Dataset"):
        hlp.check_dem(DataArrayLike())
    with pytest.raises(TypeError, match="xr.Dataset"):
        topo.sx(DataArrayLike(), 0, 500.0)
    with pytest.raises(TypeError):
        topo.sx(np.zeros((4, 5), np.float32), 0, 500.0)
