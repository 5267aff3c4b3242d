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
    # the wrapper standardises with the float64 mean / std formed on the GPU, topo.valley_ridge with
    # numpy's float32 ones like the reference: the same to the last float32 bit or nearly so
    assert np.max(np.abs(got[mask] - want[0][mask])) <= 1e-5 * np.max(want[0])
    assert np.mean(out["valley_DIR_200M"][mask] == want[1][mask]) >= 0.999
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
