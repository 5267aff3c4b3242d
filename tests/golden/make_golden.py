#!/usr/bin/env python3
"""Generate the golden vectors in this directory by RUNNING THE REAL REFERENCE.

Run once, in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference (MeteoSwiss/topo-descriptors, ``/root/reference/topo_descriptors``) is
imported unmodified.  Five of its dependencies are absent from this image
(yaconfigobject, xarray, dask, numba, utm); none of them carries hot-path arithmetic, so
they are replaced by inert stand-ins registered in ``sys.modules`` before the import
(SURVEY.md section 8c).  ``numba.njit`` becomes the identity decorator, so
``_sx_rolling`` runs as plain Python: Sx fixtures are kept small.

What is stored per fixture: the inputs, the reference outputs, and for the noisy outputs
the float64 "exact" evaluation of the same formula (from ``oracle/topo_oracle.py``) so that
the reference's own noise floor is recorded next to the numbers (SURVEY.md section 8,
tolerance contract).  Only data is written; no reference source text is stored.
"""

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


def _install_shims():
    class _Cfg:
        min_elevation = -100
        scale_std = 4

    yac = types.ModuleType("yaconfigobject")
    yac.Config = lambda name=None: _Cfg()
    sys.modules["yaconfigobject"] = yac

    xr = types.ModuleType("xarray")

    class Dataset:  # only used for isinstance checks by the reference
        pass

    class DataArray:
        pass

    xr.Dataset = Dataset
    xr.DataArray = DataArray
    sys.modules["xarray"] = xr

    dask = types.ModuleType("dask")
    dask_array = types.ModuleType("dask.array")

    class Array:
        pass

    dask_array.Array = Array
    dask.array = dask_array
    sys.modules["dask"] = dask
    sys.modules["dask.array"] = dask_array

    numba = types.ModuleType("numba")

    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f

    numba.njit = njit
    numba.prange = range
    sys.modules["numba"] = numba
    sys.modules["utm"] = types.ModuleType("utm")
    return xr


XR = _install_shims()
sys.path.insert(0, "/root/reference")
from topo_descriptors import topo as ref_topo  # noqa: E402
from topo_descriptors import helpers as ref_hlp  # noqa: E402

from oracle import topo_oracle as orc  # noqa: E402


class _Var:
    def __init__(self, values, dims):
        self.values = values
        self.dims = dims
        self.data = values


class FakeDataset(XR.Dataset):
    """Just enough of an xarray.Dataset for the reference's sx / scale_to_pixel."""

    def __init__(self, dem, x, y, crs="epsg:2056"):
        self._vars = {"dem": _Var(dem, ("y", "x")), "x": _Var(x, ("x",)), "y": _Var(y, ("y",))}
        self.attrs = {"crs": crs}

    def __getitem__(self, key):
        return self._vars[key]

    def __iter__(self):
        return iter(["dem"])


def grid(ny, nx, dx=30.0, dy=-30.0, x0=2600000.0, y0=1200000.0):
    return x0 + dx * np.arange(nx, dtype=np.float64), y0 + dy * np.arange(ny, dtype=np.float64)


def save(name, **arrays):
    """Write one fixture file.  ``<key>_exact`` arrays are not stored (the oracle travels
    and recomputes them); only the reference's noise floor max|ref - exact| is kept, as
    ``<key>_floor``."""
    path = os.path.join(HERE, name + ".npz")
    for key in [k for k in arrays if k.endswith("_exact")]:
        exact = arrays.pop(key)
        ref = arrays[key[: -len("_exact")]]
        if "aspect" in key:
            diff = orc.wrapped_angle_diff(ref, exact)
        else:
            diff = np.abs(np.asarray(ref, dtype=np.float64) - exact)
        arrays[key[: -len("_exact")] + "_floor"] = np.array(np.nanmax(diff))
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    dem_int = orc.synthetic_dem(128, 160, seed=1, integer=True)
    dem_frac = orc.synthetic_dem(96, 112, seed=2, integer=False)
    dem_big = orc.synthetic_dem(300, 280, seed=3, integer=True)

    # ---- helpers -------------------------------------------------------------------
    odd_in = np.arange(0.1, 10, 0.7)
    x30, y30 = grid(40, 50, 30.0, -30.0)
    x25, y25 = grid(40, 50, 25.0, -25.0)
    ds30 = FakeDataset(np.zeros((40, 50), np.float32), x30, y30)
    ds25 = FakeDataset(np.zeros((40, 50), np.float32), x25, y25)
    px30, res30 = ref_hlp.scale_to_pixel([2000, 200, 500], ds30)
    px25, res25 = ref_hlp.scale_to_pixel([2000, 200, 500], ds25)
    sig = ref_hlp.get_sigmas([None, 0.5, 1, 0], np.array([67, 7, 17, 9]))
    save(
        "helpers",
        odd_in=odd_in,
        odd_out=ref_hlp.round_up_to_odd(odd_in),
        x30=x30, y30=y30, px30=px30, res30_x=res30["x"], res30_y=res30["y"],
        x25=x25, y25=y25, px25=px25, res25_x=res25["x"], res25_y=res25["y"],
        sigmas=np.array([np.nan if s is None else s for s in sig], dtype=np.float64),
    )

    # ---- disc kernels --------------------------------------------------------------
    ker = {f"k{s}": ref_topo.circular_kernel(s) for s in (1, 2, 3, 4, 5, 6, 7, 8, 9, 17, 65, 67)}
    save("circular_kernel", **ker)

    # ---- TPI / STD -----------------------------------------------------------------
    out = {"dem_int": dem_int, "dem_frac": dem_frac}
    for tag, dem in (("int", dem_int), ("frac", dem_frac)):
        for size in (3, 5, 6, 7, 17, 65, 67):  # 67 px = 2000 m at 30 m: the headline size
            out[f"tpi_{tag}_s{size}"] = ref_topo.tpi(dem, size)
            out[f"tpi_{tag}_s{size}_exact"] = orc.tpi_exact(dem, size)
            out[f"std_{tag}_s{size}"] = ref_topo.std(dem, size)
            out[f"std_{tag}_s{size}_exact"] = orc.std_exact(dem, size)
        out[f"tpi_{tag}_s7_sig1p75"] = ref_topo.tpi(dem, 7, sigma=1.75)
        out[f"tpi_{tag}_s7_sig1p75_exact"] = orc.tpi_exact(dem, 7, sigma=1.75)
        out[f"std_{tag}_s17_sig2p125"] = ref_topo.std(dem, 17, sigma=2.125)
        out[f"std_{tag}_s17_sig2p125_exact"] = orc.std_exact(dem, 17, sigma=2.125)
    save("tpi_std", **out)

    # ---- Gaussian primitive --------------------------------------------------------
    out = {"dem_int": dem_int, "dem_big": dem_big}
    for s in (0.75, 2.25, 3.25):
        out[f"gauss_int_{s}"] = ref_topo.dem(dem_int, s)
        out[f"gauss_int_{s}_exact"] = orc.gaussian_exact(dem_int, s)
    out["gauss_big_30.25"] = ref_topo.dem(dem_big, 30.25)
    out["gauss_big_30.25_exact"] = orc.gaussian_exact(dem_big, 30.25)
    # radius larger than the array on one axis: multiple reflections
    small = orc.synthetic_dem(20, 48, seed=4)
    out["dem_small"] = small
    out["gauss_small_8.0"] = ref_topo.dem(small, 8.0)
    out["gauss_small_8.0_exact"] = orc.gaussian_exact(small, 8.0)
    save("gaussian", **out)

    # ---- gradient / sobel ----------------------------------------------------------
    out = {"dem_int": dem_int, "dem_big": dem_big}
    ny, nx = dem_int.shape
    xn, yn = grid(ny, nx, 30.0, -30.0)   # north-up
    xs, ys = grid(ny, nx, 30.0, 30.0)    # south-up
    res_n = orc.grid_resolution(xn, yn)
    res_s = orc.grid_resolution(xs, ys)
    # a 2-D (per-pixel) resolution pair, as the WGS84 branch would hand over
    jj, ii = np.mgrid[:ny, :nx]
    res_2d = {"x": 30.0 + 0.01 * jj + 0.002 * ii, "y": -(30.0 + 0.003 * jj - 0.001 * ii)}
    out.update(res_n_x=res_n["x"], res_n_y=res_n["y"], res_s_x=res_s["x"], res_s_y=res_s["y"],
               res_2d_x=res_2d["x"], res_2d_y=res_2d["y"])
    sdx, sdy = ref_topo.sobel(dem_int)
    out["sobel_dx"], out["sobel_dy"] = sdx, sdy
    cases = [("sob_n", 0.75, res_n, 1), ("g3_n", 3.25, res_n, 1), ("g3_s", 3.25, res_s, 1),
             ("g3_2d", 3.25, res_2d, 1), ("g3_r2_n", 3.25, res_n, 2), ("g2_r05_n", 2.25, res_n, 0.5)]
    for tag, sigma, res, ratio in cases:
        got = ref_topo.gradient(dem_int, sigma, res, sig_ratio=ratio)
        exact = orc.gradient_exact(dem_int, sigma, res, sig_ratio=ratio)
        for nm, a, e in zip(("dx", "dy", "slope", "aspect"), got, exact):
            out[f"{tag}_{nm}"] = a
            out[f"{tag}_{nm}_exact"] = e
    nyb, nxb = dem_big.shape
    xb, yb = grid(nyb, nxb, 30.0, -30.0)
    res_b = orc.grid_resolution(xb, yb)
    out.update(res_b_x=res_b["x"], res_b_y=res_b["y"])
    got = ref_topo.gradient(dem_big, 30.25, res_b)
    exact = orc.gradient_exact(dem_big, 30.25, res_b)
    for nm, a, e in zip(("dx", "dy", "slope", "aspect"), got, exact):
        out[f"g30_big_{nm}"] = a
        out[f"g30_big_{nm}_exact"] = e
    # flat field and pure planes pin the aspect conventions
    flat = np.full((16, 24), 1234.0, dtype=np.float32)
    xf, yf = grid(16, 24)
    res_f = orc.grid_resolution(xf, yf)
    north_facing = (1000.0 + 3.0 * np.arange(16, dtype=np.float32))[:, None] * np.ones((1, 24), np.float32)
    east_facing = np.ones((16, 1), np.float32) * (1000.0 - 2.0 * np.arange(24, dtype=np.float32))[None, :]
    for tag, arr in (("flat", flat), ("northf", north_facing), ("eastf", east_facing)):
        got = ref_topo.gradient(np.ascontiguousarray(arr), 2.0, res_f)
        out[f"plane_{tag}_in"] = np.ascontiguousarray(arr)
        for nm, a in zip(("dx", "dy", "slope", "aspect"), got):
            out[f"plane_{tag}_{nm}"] = a
    out.update(res_f_x=res_f["x"], res_f_y=res_f["y"])
    save("gradient", **out)

    # ---- Sx geometry helpers -------------------------------------------------------
    out = {}
    geo_cases = [(500.0, 30.0, -30.0), (2000.0, 30.0, -30.0), (500.0, 25.0, -25.0),
                 (150.0, 50.0, 40.0), (300.0, 20.0, -30.0)]
    for n, (radius, dx, dy) in enumerate(geo_cases):
        out[f"dist{n}_args"] = np.array([radius, dx, dy])
        out[f"dist{n}"] = ref_topo._sx_distance(radius, dx, dy)
    az_sets = [np.linspace(-5, 5, 15), np.linspace(85, 95, 15), np.linspace(220, 230, 15),
               np.array([3.0, 4.0, 5.0, 6.0]), np.array([180.0])]
    for n, az in enumerate(az_sets):
        for m, (radius, dx, dy) in enumerate(geo_cases):
            delta = ref_topo._sx_source_idx_delta(az, radius, dx, dy)
            out[f"delta_a{n}_g{m}"] = delta
            dist = ref_topo._sx_distance(radius, dx, dy)
            centre = np.floor(np.array(dist.shape) / 2)
            source = (centre + delta).astype(int)
            out[f"lines_a{n}_g{m}"] = ref_topo._sx_bresenhamlines(source, centre)
        out[f"az{n}"] = az
    out["bres_start"] = np.array([[8, 9], [17, 22]])
    out["bres_end"] = np.array([15, 15])
    out["bres_out"] = ref_topo._sx_bresenhamlines(out["bres_start"], out["bres_end"])
    save("sx_geometry", **out)

    # ---- Sx end to end (plain-Python reference loop: keep small) ------------------
    out = {}
    dem_sx = orc.synthetic_dem(96, 104, seed=5, integer=True)
    out["dem"] = dem_sx
    ny, nx = dem_sx.shape
    sx_cases = [
        ("az0", dict(azimuth=0, radius=500.0), (30.0, -30.0)),
        ("az90", dict(azimuth=90, radius=500.0), (30.0, -30.0)),
        ("az225", dict(azimuth=225, radius=500.0), (30.0, -30.0)),
        ("arc0", dict(azimuth=45, radius=500.0, azimuth_arc=0), (30.0, -30.0)),
        ("rmin", dict(azimuth=300, radius=600.0, radius_min=200.0, height=2.0), (30.0, -30.0)),
        ("south_up", dict(azimuth=10, radius=400.0), (30.0, 30.0)),
        ("aniso", dict(azimuth=135, radius=500.0, azimuth_steps=7, azimuth_arc=20.0), (25.0, -40.0)),
    ]
    for tag, kw, (dx, dy) in sx_cases:
        x, y = grid(ny, nx, dx, dy)
        ds = FakeDataset(dem_sx, x, y)
        res = ref_topo.sx(ds, **kw)
        out[f"{tag}_out"] = res
        out[f"{tag}_x"] = x
        out[f"{tag}_y"] = y
        full = dict(height=10.0, azimuth_arc=10.0, azimuth_steps=15, radius_min=0.0)
        full.update(kw)
        out[f"{tag}_params"] = np.array([full["azimuth"], full["radius"], full["height"],
                                         full["azimuth_arc"], full["azimuth_steps"],
                                         full["radius_min"]], dtype=np.float64)
    save("sx", **out)

    # ---- valley / ridge index ------------------------------------------------------
    # Small DEMs: the reference evaluates 180 FFT convolutions per call.  The kernels the
    # reference builds (before and after rotation) are stored too, so that the oracle's and the
    # product's own constructions can be checked bit for bit without the reference at hand.
    vr_int = orc.synthetic_dem(72, 88, seed=4, integer=True)
    vr_frac = orc.synthetic_dem(64, 80, seed=5, integer=False)
    out = {"dem_int": vr_int, "dem_frac": vr_frac}
    cases = [
        ("int_valley_s7", vr_int, 7, "valley", [0, 0.15, 0.3], None),
        ("int_ridge_s7", vr_int, 7, "ridge", [0, 0.15, 0.3], None),
        ("int_valley_s5", vr_int, 5, "valley", [0, 0.15, 0.3], None),
        ("int_valley_s17", vr_int, 17, "valley", [0, 0.15, 0.3], None),
        ("int_valley_s9_flat0", vr_int, 9, "valley", [0], None),
        ("int_ridge_s9_flat2", vr_int, 9, "ridge", [0.2, 0.4], None),
        ("frac_valley_s7", vr_frac, 7, "valley", [0, 0.15, 0.3], None),
        ("frac_valley_s9_sig", vr_frac, 9, "valley", [0, 0.15, 0.3], 1.125),
    ]
    for tag, dem, size, mode, flats, sigma in cases:
        norm, direction = ref_topo.valley_ridge(dem, size, mode, flats, sigma)
        exact = orc.valley_ridge_exact(dem, size, mode, flats, sigma)
        out[f"{tag}_norm"] = norm
        out[f"{tag}_norm_exact"] = exact[0]
        out[f"{tag}_dir"] = direction
        out[f"{tag}_params"] = np.array([size, 0 if mode == "valley" else 1, -1.0 if sigma is None else sigma]
                                        + list(flats), dtype=np.float64)
    for size, flats in ((5, [0, 0.15, 0.3]), (7, [0, 0.15, 0.3]), (9, [0.2, 0.4]), (17, [0, 0.15, 0.3])):
        base = ref_topo._valley_kernels(size, flats)
        out[f"kernels_s{size}_n{len(flats)}"] = base
        for angle in (0, 1, 33, 45, 90, 137, 179):
            out[f"kernels_s{size}_n{len(flats)}_rot{angle}"] = ref_topo._rotate_kernels(base, np.float32(angle))
    save("valley_ridge", **out)

    # ---- batch wrappers: the reference's compute_* with its netCDF writer captured -------
    # compute_dem / compute_tpi / compute_std / compute_gradient / compute_sx / compute_valley_ridge
    # (topo.py:16-59, 88-141, 216-269, 534-594, 715-772, 317-386) run unmodified on a Dataset-like DEM;
    # helpers.to_netcdf (helpers.py:34-65) is replaced by a function that keeps {NAME: array}, so the
    # fixture pins the scale -> pixel -> sigma plumbing, the output names and the NaN re-insertion.
    dem_b = orc.synthetic_dem(96, 112, seed=6, integer=True)
    nyb, nxb = dem_b.shape
    xb, yb = grid(nyb, nxb, 30.0, -30.0)
    ds_b = FakeDataset(dem_b, xb, yb)
    ind_nans = (np.array([3, 50, 95]), np.array([4, 60, 111]))
    captured = {}

    def capture(array, dem_ds, name, crop=None, outdir=".", units=None):
        captured[str.upper(name)] = (np.array(array, copy=True), units)

    ref_hlp.to_netcdf = capture
    out = {"dem": dem_b, "x": xb, "y": yb, "nan_rows": ind_nans[0], "nan_cols": ind_nans[1]}
    px_b, res_b2 = ref_hlp.scale_to_pixel([100, 200, 400, 500], ds_b)
    out["px_100_200_400_500"] = px_b
    calls = [
        ("tpi", lambda: ref_topo.compute_tpi(ds_b, [200, 500], smth_factors=[None, 0.5], ind_nans=ind_nans)),
        ("std", lambda: ref_topo.compute_std(ds_b, [200, 500], smth_factors=0.5, ind_nans=ind_nans)),
        ("std0", lambda: ref_topo.compute_std(ds_b, 200, ind_nans=ind_nans)),
        ("grad", lambda: ref_topo.compute_gradient(ds_b, [100, 400], sig_ratios=[1, 2], ind_nans=ind_nans)),
        ("dem", lambda: ref_topo.compute_dem(ds_b, [400], ind_nans=ind_nans)),
        ("sx", lambda: ref_topo.compute_sx(ds_b, 0, 300.0)),
        ("sx225", lambda: ref_topo.compute_sx(ds_b, 225, 300.0, height=2.0, azimuth_arc=20.0, azimuth_steps=7)),
        ("vr", lambda: ref_topo.compute_valley_ridge(ds_b, [200], "valley", smth_factors=[None], ind_nans=ind_nans)),
    ]
    names = []
    for tag, fn in calls:
        captured.clear()
        fn()
        for name, (array, units) in captured.items():
            out[f"{tag}__{name}"] = array
            names.append(f"{tag}__{name}|{units}")
    out["names_units"] = np.array(names)
    # float64 evaluations of the same formulas: the reference's noise floor travels with the fixture
    sig17 = 0.5 * 17 / 4
    out["tpi__TPI_200M_exact"] = orc.tpi_exact(dem_b, 7)
    out["tpi__TPI_500M_SMTHFACT0.5_exact"] = orc.tpi_exact(dem_b, 17, sigma=sig17)
    out["std__STD_200M_SMTHFACT0.5_exact"] = orc.std_exact(dem_b, 7, sigma=0.5 * 7 / 4)
    out["std__STD_500M_SMTHFACT0.5_exact"] = orc.std_exact(dem_b, 17, sigma=sig17)
    out["std0__STD_200M_exact"] = orc.std_exact(dem_b, 7)
    for key in [k for k in out if k.endswith("_exact")]:
        ref = out[key[: -len("_exact")]]
        exact = np.array(out[key], dtype=np.float64)
        exact[ind_nans] = np.nan  # the wrappers put the NaNs back
        out[key] = exact
        assert np.isnan(ref[ind_nans]).all()
    save("batch", **out)


if __name__ == "__main__":
    main()
