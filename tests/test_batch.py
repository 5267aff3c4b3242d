"""Batch wrappers: names on CPU, results on the GPU."""
import numpy as np
import pytest

from topo_descriptors_amd import batch


def test_output_names_match_the_reference_strings():
    # reference topo.py:83, :184-188, :310-314, :647-655, :956-960
    assert batch._dem_name(500) == "DEM_500M"
    assert batch._tpi_name(2000, None) == "TPI_2000M"
    assert batch._tpi_name(2000, 0.5) == "TPI_2000M_SMTHFACT0.5"
    assert batch._std_name(200, 1) == "STD_200M_SMTHFACT1"
    assert batch._gradient_names(500, 1) == ["WE_DERIVATIVE_500M_SIGRATIO1", "SN_DERIVATIVE_500M_SIGRATIO1",
                                             "SLOPE_500M_SIGRATIO1", "ASPECT_500M_SIGRATIO1"]
    assert batch._gradient_names(500, 0.25)[2] == "SLOPE_500M_SIGRATIO0.25"
    assert batch._sx_name(500.0, 225.7) == "SX_RADIUS500_AZIMUTH225"
    assert batch._valley_ridge_names(2000, "valley", None) == ["valley_NORM_2000M", "valley_DIR_2000M"]
    assert batch._valley_ridge_names(500, "ridge", 0.5) == ["ridge_NORM_500M_SMTHFACT0.5", "ridge_DIR_500M_SMTHFACT0.5"]


def test_the_wrappers_are_reachable_from_topo_like_in_the_reference():
    from topo_descriptors_amd import topo
    for name in ("compute_dem", "compute_tpi", "compute_std", "compute_gradient", "compute_sx",
                 "compute_valley_ridge"):
        assert getattr(topo, name) is getattr(batch, name)
    with pytest.raises(AttributeError):
        topo.compute_nothing  # noqa: B018


class FakeVar:
    def __init__(self, values, dims):
        self.values, self.dims = values, dims


class FakeDataset:
    def __init__(self, dem, x, y):
        self._v = {"dem": FakeVar(dem, ("y", "x")), "x": FakeVar(x, ("x",)), "y": FakeVar(y, ("y",))}
        self.attrs = {"crs": "epsg:2056"}

    def __getitem__(self, k):
        return self._v[k]

    def __iter__(self):
        return iter(["dem"])


@pytest.mark.gpu
def test_compute_tpi_pairs_small_scales_in_one_pass():
    """Un-smoothed TPI at small scales goes two sizes per pass over the DEM (Block.tpi_multi); the planes, their
    names and the NaN re-insertion are those of the scale-by-scale loop (reference topo.py:132-141)."""
    from oracle import topo_oracle as orc
    from topo_descriptors_amd import helpers as hlp, topo

    ny, nx = 200, 256
    dem = orc.synthetic_dem(ny, nx, seed=11, integer=False)
    x = 2600000.0 + 30.0 * np.arange(nx)
    y = 1200000.0 - 30.0 * np.arange(ny)
    ds = FakeDataset(dem, x, y)
    scales = [150, 200, 500, 260, 330, 200]        # 5, 7, 17, 9, 11 px, and 7 px once more
    px, _ = hlp.scale_to_pixel(scales, ds)
    assert list(px) == [5, 7, 17, 9, 11, 7]
    ind_nans = (np.array([3, 50]), np.array([4, 60]))
    out = batch.compute_tpi(ds, scales, smth_factors=[None, None, None, None, None, 1], ind_nans=ind_nans, outdir=None)
    assert set(out) == {"TPI_150M", "TPI_200M", "TPI_500M", "TPI_260M", "TPI_330M", "TPI_200M_SMTHFACT1"}
    for scale, size in zip(scales[:5], px[:5]):
        got, want = out[f"TPI_{scale}M"], topo.tpi(dem, int(size))
        assert np.isnan(got[3, 4]) and np.isnan(got[50, 60])
        want[ind_nans] = np.nan
        assert np.array_equal(got, want, equal_nan=True), scale
    want = topo.tpi(dem, 7, sigma=7 / 4)
    want[ind_nans] = np.nan
    assert np.array_equal(out["TPI_200M_SMTHFACT1"], want, equal_nan=True)


@pytest.mark.gpu
def test_wrappers_equal_the_single_calls(tmp_path):
    from oracle import topo_oracle as orc
    from topo_descriptors_amd import helpers as hlp, topo

    ny, nx = 180, 256
    dem = orc.synthetic_dem(ny, nx, seed=8)
    dem[5, 7] = np.nan
    x = 2600000.0 + 30.0 * np.arange(nx)
    y = 1200000.0 - 30.0 * np.arange(ny)
    filled = np.where(np.isnan(dem), 1900.0, dem).astype(np.float32)
    ind_nans = np.where(np.isnan(dem))
    ds = FakeDataset(filled, x, y)
    scales = [200, 500]
    px, res = hlp.scale_to_pixel(scales, ds)
    assert list(px) == [7, 17]

    out = batch.compute_tpi(ds, scales, smth_factors=[None, 0.5], ind_nans=ind_nans, outdir=str(tmp_path))
    assert set(out) == {"TPI_200M", "TPI_500M_SMTHFACT0.5"}
    want = topo.tpi(filled, 7)
    got = out["TPI_200M"]
    assert np.isnan(got[5, 7])
    mask = ~np.isnan(got)
    assert np.array_equal(got[mask], want[mask])
    assert np.array_equal(np.load(tmp_path / "topo_TPI_200M.npy")[mask], want[mask])
    want = topo.tpi(filled, 17, sigma=0.5 * 17 / 4)
    assert np.array_equal(out["TPI_500M_SMTHFACT0.5"][mask], want[mask])

    out = batch.compute_std(ds, 200, outdir=None)
    assert out["STD_200M"].dtype == np.float64
    assert np.array_equal(out["STD_200M"], topo.std(filled, 7))

    out = batch.compute_gradient(ds, scales, outdir=None)
    want = topo.gradient(filled, 17 / 4, res)
    for k, name in enumerate(batch._gradient_names(500, 1)):
        assert np.array_equal(out[name], want[k]), name

    out = batch.compute_dem(ds, 500, outdir=None)
    assert np.array_equal(out["DEM_500M"], topo.dem(filled, 17 / 4))

    out = batch.compute_sx(ds, 0, 300.0, outdir=None)
    assert np.array_equal(out["SX_RADIUS300_AZIMUTH0"], topo.sx(ds, 0, 300.0))
    fan = batch.compute_sx(ds, [0, 5, 90], 300.0, outdir=None)
    assert sorted(fan) == ["SX_RADIUS300_AZIMUTH0", "SX_RADIUS300_AZIMUTH5", "SX_RADIUS300_AZIMUTH90"]
    for az in (0, 5, 90):
        assert np.array_equal(fan[f"SX_RADIUS300_AZIMUTH{az}"], topo.sx(ds, az, 300.0))

    out = batch.compute_valley_ridge(ds, 200, "valley", smth_factors=None, ind_nans=ind_nans, outdir=None)
    assert set(out) == {"valley_NORM_200M", "valley_DIR_200M"}
    want = topo.valley_ridge(filled, 7, "valley")
    got = out["valley_NORM_200M"]
    assert np.isnan(got[5, 7]) and np.isnan(out["valley_DIR_200M"][5, 7])
    # the wrapper and topo.valley_ridge standardise with the same numbers (numpy's float32 mean / std, like
    # the reference): identical bits
    assert np.array_equal(got[mask], want[0][mask])
    assert np.array_equal(out["valley_DIR_200M"][mask], want[1][mask])
    with pytest.raises(ValueError):
        batch.compute_valley_ridge(ds, 200, "canyon", outdir=None)


@pytest.mark.gpu
def test_the_calls_of_the_reference_example_script_at_its_large_scales():
    """scripts/compute_topo_descriptors.py of the reference, call by call, on a 100 m grid with scales
    up to 30 km: disc sizes 1 ... 301 px (larger than the wave-shift kernels cover), Gaussian sigma up
    to 75, valley / ridge kernels up to ~430 px (the FFT route).  Every output must carry the
    reference's name and equal the single call it stands for (whose parity with the oracle the
    other GPU tests hold, test_very_large_discs and test_kernels_beyond_the_lds_tile... included)."""
    from oracle import topo_oracle as orc
    from topo_descriptors_amd import helpers as hlp, topo

    ny, nx = 330, 400
    dem = orc.synthetic_dem(ny, nx, seed=12)
    x = 2600000.0 + 100.0 * np.arange(nx)
    y = 1200000.0 - 100.0 * np.arange(ny)
    ds = FakeDataset(dem, x, y)
    scales = [100, 500, 2000, 10000, 30000]
    px, res = hlp.scale_to_pixel(scales, ds)
    assert list(px) == [1, 5, 21, 101, 301]
    ind_nans = (np.array([3, 200]), np.array([4, 17]))

    def same(got, want, name):
        mask = np.ones(got.shape, bool)
        mask[ind_nans] = False
        assert np.isnan(got[ind_nans]).all(), name
        # (a 1-pixel disc has no neighbour: 0 / 0 everywhere, in the reference as well)
        assert np.array_equal(got[mask], np.asarray(want)[mask], equal_nan=True), name

    out = batch.compute_dem(ds, scales, ind_nans=ind_nans, outdir=None)
    assert sorted(out) == sorted(f"DEM_{s}M" for s in scales)
    same(out["DEM_30000M"], topo.dem(dem, 301 / 4), "DEM_30000M")
    out = batch.compute_tpi(ds, scales, smth_factors=None, ind_nans=ind_nans, outdir=None)
    for s, p in zip(scales, px):
        same(out[f"TPI_{s}M"], topo.tpi(dem, int(p)), f"TPI_{s}M")
    out = batch.compute_tpi(ds, scales, smth_factors=1, ind_nans=ind_nans, outdir=None)
    same(out["TPI_10000M_SMTHFACT1"], topo.tpi(dem, 101, sigma=101 / 4), "TPI_10000M_SMTHFACT1")
    out = batch.compute_gradient(ds, scales, sig_ratios=1, ind_nans=ind_nans, outdir=None)
    want = topo.gradient(dem, 301 / 4, res)
    for k, name in enumerate(batch._gradient_names(30000, 1)):
        same(out[name], want[k], name)
    out = batch.compute_std(ds, scales, ind_nans=ind_nans, outdir=None)
    for s, p in zip(scales, px):
        same(out[f"STD_{s}M"], topo.std(dem, int(p)), f"STD_{s}M")
    for mode, flats in (("valley", [0, 0.2, 0.4]), ("ridge", [0, 0.15, 0.3])):
        out = batch.compute_valley_ridge(ds, scales[3:], mode, flat_list=flats, smth_factors=0.5,
                                         ind_nans=ind_nans, outdir=None)
        assert sorted(out) == sorted(n for s in scales[3:] for n in batch._valley_ridge_names(s, mode, 0.5))
        for name, array in out.items():
            assert np.isnan(array[ind_nans]).all() and np.isfinite(np.delete(array.ravel(), ind_nans[0] * nx + ind_nans[1])).all(), name
        norm = out[batch._valley_ridge_names(10000, mode, 0.5)[0]]
        want = topo.valley_ridge(dem, 101, mode, flats, sigma=0.5 * 101 / 4)[0]
        mask = ~np.isnan(norm)
        assert np.max(np.abs(norm[mask] - want[mask])) <= 2e-5 * np.max(want), mode
    out = batch.compute_sx(ds, 0, 1000, outdir=None)
    assert np.array_equal(out["SX_RADIUS1000_AZIMUTH0"], topo.sx(ds, 0, 1000))


# ---- n1 pinned on the reference (task 8) --------------------------------------------
# tests/golden/batch.npz holds what the reference's own compute_dem / compute_tpi / compute_std /
# compute_gradient / compute_sx / compute_valley_ridge (topo.py:16-59, 88-141, 216-269, 534-594, 715-772,
# 317-386) handed to its netCDF writer (helpers.to_netcdf, replaced by a capture function in
# tests/golden/make_golden.py), keyed "<call>__<NAME>", plus the units string of every output.
def _batch_fixture(golden):
    g = golden("batch")
    ds = FakeDataset(g["dem"], g["x"], g["y"])
    ind_nans = (g["nan_rows"], g["nan_cols"])
    expected = {}
    for entry in g["names_units"]:
        key, units = str(entry).split("|")
        expected[key] = (g[key], units)
    return g, ds, ind_nans, expected


def _batch_calls(ds, ind_nans):
    return {
        "tpi": lambda: batch.compute_tpi(ds, [200, 500], smth_factors=[None, 0.5], ind_nans=ind_nans, outdir=None),
        "std": lambda: batch.compute_std(ds, [200, 500], smth_factors=0.5, ind_nans=ind_nans, outdir=None),
        "std0": lambda: batch.compute_std(ds, 200, ind_nans=ind_nans, outdir=None),
        "grad": lambda: batch.compute_gradient(ds, [100, 400], sig_ratios=[1, 2], ind_nans=ind_nans, outdir=None),
        "dem": lambda: batch.compute_dem(ds, [400], ind_nans=ind_nans, outdir=None),
        "sx": lambda: batch.compute_sx(ds, 0, 300.0, outdir=None),
        "sx225": lambda: batch.compute_sx(ds, 225, 300.0, height=2.0, azimuth_arc=20.0, azimuth_steps=7, outdir=None),
        "vr": lambda: batch.compute_valley_ridge(ds, [200], "valley", smth_factors=[None], ind_nans=ind_nans, outdir=None),
    }


def test_reference_batch_fixture_plumbing(golden):
    """CPU: the fixture's names, pixel sizes and sigmas are the ones this package's host logic derives, and
    the oracle's scipy evaluators at those parameters reproduce the reference's wrapper outputs bit for bit
    (NaNs put back at ind_nans)."""
    from oracle import topo_oracle as orc
    from topo_descriptors_amd import helpers as hlp

    g, ds, ind_nans, expected = _batch_fixture(golden)
    px, res = hlp.scale_to_pixel([100, 200, 400, 500], ds)
    assert list(px) == list(g["px_100_200_400_500"]) == [3, 7, 13, 17]
    sig = hlp.get_sigmas([None, 0.5], np.array([7, 17]))
    assert sig[0] is None and sig[1] == 0.5 * 17 / 4
    names = {
        "tpi": [batch._tpi_name(200, None), batch._tpi_name(500, 0.5)],
        "std": [batch._std_name(200, 0.5), batch._std_name(500, 0.5)],
        "std0": [batch._std_name(200, None)],
        "grad": batch._gradient_names(100, 1) + batch._gradient_names(400, 2),
        "dem": [batch._dem_name(400)],
        "sx": [batch._sx_name(300.0, 0)],
        "sx225": [batch._sx_name(300.0, 225)],
        "vr": batch._valley_ridge_names(200, "valley", None),
    }
    want_keys = {f"{tag}__{str.upper(n)}" for tag, ns in names.items() for n in ns}
    assert want_keys == set(expected)
    units = {k: u for k, (_, u) in expected.items()}
    assert units["tpi__TPI_200M"] == "m" and units["grad__SLOPE_100M_SIGRATIO1"] == "degree"
    assert units["grad__WE_DERIVATIVE_400M_SIGRATIO2"] == "1" and units["sx__SX_RADIUS300_AZIMUTH0"] == "degree"

    dem = g["dem"]

    def with_nans(a):
        a = np.array(a, copy=True)
        a[ind_nans] = np.nan
        return a

    assert np.array_equal(with_nans(orc.tpi_scipy(dem, 7)), expected["tpi__TPI_200M"][0], equal_nan=True)
    assert np.array_equal(with_nans(orc.tpi_scipy(dem, 17, sigma=0.5 * 17 / 4)), expected["tpi__TPI_500M_SMTHFACT0.5"][0],
                          equal_nan=True)
    assert np.array_equal(with_nans(orc.std_scipy(dem, 7, sigma=0.5 * 7 / 4)), expected["std__STD_200M_SMTHFACT0.5"][0],
                          equal_nan=True)
    assert np.array_equal(with_nans(orc.std_scipy(dem, 7)), expected["std0__STD_200M"][0], equal_nan=True)
    assert np.array_equal(with_nans(orc.gaussian_scipy(dem, 13 / 4)), expected["dem__DEM_400M"][0], equal_nan=True)
    got = orc.gradient_scipy(dem, 13 / 4, res, sig_ratio=2)
    for k, n in enumerate(batch._gradient_names(400, 2)):
        assert np.array_equal(with_nans(got[k]), expected[f"grad__{n}"][0], equal_nan=True), n
    got = orc.gradient_scipy(dem, 3 / 4, res, sig_ratio=1)  # sigma <= 1: the Sobel branch
    for k, n in enumerate(batch._gradient_names(100, 1)):
        assert np.array_equal(with_nans(got[k]), expected[f"grad__{n}"][0], equal_nan=True), n
    assert np.max(np.abs(orc.sx(dem, g["x"], g["y"], 0, 300.0) - expected["sx__SX_RADIUS300_AZIMUTH0"][0])) <= 2e-5


@pytest.mark.gpu
def test_wrappers_against_the_reference_wrappers(golden):
    """GPU: batch.compute_* against what the reference's compute_* produced on the same Dataset - names, NaN
    positions, values to the contract of SURVEY section 8 (STD two-sided with the stored reference floor)."""
    from oracle import topo_oracle as orc

    g, ds, ind_nans, expected = _batch_fixture(golden)
    got = {}
    for tag, fn in _batch_calls(ds, ind_nans).items():
        for name, array in fn().items():
            got[f"{tag}__{str.upper(name)}"] = array
    assert set(got) == set(expected)
    for key, (ref, _) in expected.items():
        out = got[key]
        assert out.shape == ref.shape and out.dtype == ref.dtype, (key, out.dtype, ref.dtype)
        assert np.array_equal(np.isnan(out), np.isnan(ref)), key
        m = ~np.isnan(ref)
        scale = np.max(np.abs(ref[m]))
        if "STD" in key:
            exact, floor = g[key + "_exact"] if key + "_exact" in g else None, float(g[key + "_floor"])
            assert np.max(np.abs(out[m] - ref[m])) <= floor + 1e-4 * scale, key
        elif "ASPECT" in key:
            slope = got[key.replace("ASPECT", "SLOPE")]
            steep = m & (slope > 0.1)
            assert np.max(orc.wrapped_angle_diff(out[steep], ref[steep])) <= 0.036, key
        elif "VALLEY_DIR" in key:
            assert np.mean(out[m] == ref[m]) >= 0.999, key
        else:
            assert np.max(np.abs(out[m] - ref[m])) <= 1e-4 * scale, (key, np.max(np.abs(out[m] - ref[m])), scale)


@pytest.mark.gpu
def test_a_wgs84_dataset_goes_through_the_batch_wrappers():
    """reference helpers.py:89-97: a DEM on a longitude / latitude grid is reprojected to UTM for its resolution; the `utm`
    package is absent here and helpers._utm_from_latlon stands in (pinned on CPU in tests/test_host_api.py).  The gradient then
    divides by per-pixel resolutions (topo.py:688-712).  Checked against the oracle fed the same projected coordinates."""
    from oracle import topo_oracle as orc
    from topo_descriptors_amd import helpers as hlp
    rng = np.random.default_rng(5)
    ny, nx = 150, 210
    x = 8.0 + np.arange(nx) / 1200.0                              # 3 arc seconds, 46.5 N: ~64 m x ~93 m
    y = 46.5 - np.arange(ny) / 1200.0
    yy, xx = np.mgrid[0:ny, 0:nx]
    dem = np.round(1500 + 400 * np.sin(xx / 23.0) * np.cos(yy / 17.0) + rng.uniform(0, 20, (ny, nx))).astype(np.float32)
    ds = FakeDataset(dem, x, y)
    ds.attrs["crs"] = "EPSG:4326"
    px, res = hlp.scale_to_pixel([500, 1000], ds)
    assert res["x"].ndim == 2 and res["y"].ndim == 2 and list(px) == [7, 13]
    east, north = hlp._wgs84_to_utm(x, y)
    opx, ores = orc.scale_to_pixel([500, 1000], east, north)
    assert np.array_equal(px, opx) and np.array_equal(res["x"], ores["x"]) and np.array_equal(res["y"], ores["y"])

    out = batch.compute_gradient(ds, [500, 1000], outdir=None)
    for scale, p in zip((500, 1000), px):
        want = orc.gradient_scipy(dem, p / 4, ores)
        names = batch._gradient_names(scale, 1)
        for k in range(3):
            assert np.max(np.abs(out[names[k]] - want[k])) <= 1e-4 * np.max(np.abs(want[k])), names[k]
        steep = want[2] > 0.1
        assert np.max(orc.wrapped_angle_diff(out[names[3]], want[3])[steep]) <= 1e-4 * 360.0
    out = batch.compute_tpi(ds, [500, 1000], outdir=None)
    for scale, p in zip((500, 1000), px):
        want = orc.tpi_exact(dem, int(p))
        assert np.max(np.abs(out[f"TPI_{scale}M"] - want)) <= 1e-4 * np.max(np.abs(want))
