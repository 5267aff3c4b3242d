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
