"""Host-side parameter preparation: metres -> pixels, sigmas, DEM validation.

Mirrors the call signatures of the reference's ``topo_descriptors/helpers.py`` for the part of
it the descriptor hot path uses (scale_to_pixel :68, round_up_to_odd :108, get_sigmas :114,
check_dem :171, get_da :191).  No per-pixel arithmetic happens here.  xarray is optional:
anything that quacks like a Dataset (``ds["x"].values``, ``ds.attrs["crs"]``, iterable
variable names, ``ds[name].dims``) is accepted, because the GPU box may not have xarray.
"""
import datetime
import functools
import logging
import time

import numpy as np

from . import CFG

logger = logging.getLogger(__name__)

try:  # pragma: no cover - xarray is absent from the build image
    import xarray as _xr
except Exception:  # noqa: BLE001
    _xr = None


def _looks_like_dataset(obj):
    if _xr is not None and isinstance(obj, _xr.Dataset):
        return True
    if isinstance(obj, np.ndarray):
        return False
    # a DataArray has attrs, __getitem__ and __iter__ too; the reference raises for it (topo.py:825-826,
    # helpers.py:179-180: isinstance(dem, xr.Dataset)).  It carries ``dims`` itself and has no data variables.
    if hasattr(obj, "dims") and not hasattr(obj, "data_vars"):
        return False
    return all(hasattr(obj, a) for a in ("attrs", "__getitem__", "__iter__"))


def check_dem(dem):
    """Validate the DEM data model (reference helpers.py:171-188): Dataset, dims ('y','x'),
    a ``crs`` attribute naming an EPSG code."""
    if not _looks_like_dataset(dem):
        raise ValueError("dem must be a xr.Dataset")
    first = list(dem)[0]
    if tuple(dem[first].dims) != ("y", "x"):
        raise ValueError("dem dimensions must be ('y', 'x')")
    if "crs" not in dem.attrs:
        raise KeyError("missing 'crs' (case sensitive) attribute in dem")
    if "epsg:" not in dem.attrs["crs"].lower():
        raise ValueError("missing 'epsg:' (case insensitive) key in the 'crs' attribute")


def get_da(dem_ds):
    """First data variable of the Dataset, whatever its name (reference helpers.py:191-196)."""
    return dem_ds[list(dem_ds)[0]]


def round_up_to_odd(f):
    """Nearest odd integer as int64 (reference helpers.py:108-111)."""
    f = np.asarray(f, dtype=np.float64)
    return (np.round((f - 1.0) / 2.0) * 2.0 + 1.0).astype(np.int64)


def _wgs84_to_utm(x_coords, y_coords):
    try:
        import utm  # noqa: PLC0415
    except Exception as exc:  # noqa: BLE001
        raise RuntimeError("WGS84 (epsg:4326) grids need the 'utm' package to derive the "
                           "resolution in metres") from exc
    lon, lat = np.meshgrid(x_coords, y_coords)
    east, north, _, _ = utm.from_latlon(lat, lon)
    return east.astype(np.float32), north.astype(np.float32)


def scale_to_pixel(scales, dem_ds):
    """Scales in metres -> odd pixel diameters, plus the signed per-node grid resolution
    ``{"x": ..., "y": ...}`` in metres (reference helpers.py:68-105)."""
    check_dem(dem_ds)
    x_coords = np.asarray(dem_ds["x"].values)
    y_coords = np.asarray(dem_ds["y"].values)
    if "epsg:4326" in dem_ds.attrs["crs"].lower():
        logger.debug("Reprojecting coordinates from WGS84 to UTM to obtain units of meters")
        x_coords, y_coords = _wgs84_to_utm(x_coords, y_coords)
    x_res = np.gradient(x_coords, axis=x_coords.ndim - 1)
    y_res = np.gradient(y_coords, axis=0)
    mean_res = np.mean(np.abs([x_res.mean(), y_res.mean()]))
    logger.debug("Estimated resolution: %.0f meters.", mean_res)
    return round_up_to_odd(np.array(scales) / mean_res), {"x": x_res, "y": y_res}


def get_sigmas(smth_factors, scales_pxl):
    """Gaussian sigmas in pixels for smoothing factors; falsy factor -> None
    (reference helpers.py:114-134)."""
    factors = np.array([f if f else np.nan for f in smth_factors], dtype=np.float64)
    sigmas = factors * np.asarray(scales_pxl) / CFG.scale_std
    return [None if np.isnan(s) else s for s in sigmas]


# ---- the steps either side of the path: thin, and xarray's where the reference uses it
def _need_xarray(what):
    if _xr is None:
        raise ImportError(f"{what} reads / writes netCDF through xarray, which is not installed here; "
                          "the descriptor functions themselves accept any Dataset-like object")
    return _xr


def get_dem_netcdf(path_dem):
    """The DEM of a netCDF file as a float32 Dataset, elevations at or below ``CFG.min_elevation``
    masked as NaN (reference helpers.py:17-31).  Needs xarray."""
    xr = _need_xarray("get_dem_netcdf")
    dem_ds = xr.open_dataset(path_dem).astype(np.float32).squeeze(drop=True)
    return dem_ds.where(dem_ds > CFG.min_elevation)


def to_netcdf(array, dem_ds, name, crop=None, outdir=".", units=None):
    """``array`` saved as ``topo_<NAME>.nc`` with the coordinates and attributes of ``dem_ds``
    (reference helpers.py:34-65).  Needs xarray; ``batch.write_output`` is the same writer and falls
    back to ``.npy`` without it."""
    _need_xarray("to_netcdf")
    from . import batch  # noqa: PLC0415  (batch imports this module)
    return batch.write_output(array, dem_ds, name, crop=crop, outdir=outdir, units=units)


def fill_na_array(values, x_coords=None):
    """NaNs of a 2-D array replaced row by row with the nearest valid sample along x, the edge value
    beyond the first / last valid sample: what ``interpolate_na(dim="x", method="nearest",
    fill_value="extrapolate")`` does in the reference's ``fill_na`` (helpers.py:137-154), through the
    same ``scipy.interpolate.interp1d`` xarray uses (a sample half way between two valid ones takes the
    left one).  Rows with fewer than two valid samples are left alone.  xarray is absent from the image, so
    this is pinned by hand-derived known answers (tests/test_host_api.py::test_fill_na_array_known_answers),
    not by a run of the reference."""
    from scipy.interpolate import interp1d  # noqa: PLC0415

    out = np.array(values, copy=True)
    x = np.arange(out.shape[1], dtype=np.float64) if x_coords is None else np.asarray(x_coords, dtype=np.float64)
    for row in out:
        bad = np.isnan(row)
        if bad.any() and (~bad).sum() >= 2:
            fill = interp1d(x[~bad], row[~bad], kind="nearest", fill_value="extrapolate", assume_sorted=False)
            row[bad] = fill(x[bad])
    return out


def fill_na(dem_ds):
    """``(ind_nans, filled Dataset)``: where the DEM has NaNs, and the DEM with them interpolated
    along x by the nearest valid value (reference helpers.py:137-154).  Needs xarray."""
    _need_xarray("fill_na")
    ind_nans = np.where(np.isnan(get_da(dem_ds)))
    return ind_nans, dem_ds.interpolate_na(dim="x", method="nearest", fill_value="extrapolate")


def timer(func):
    """Decorator that logs how long ``func`` took (reference helpers.py:157-168)."""
    @functools.wraps(func)
    def timed(*args, **kwargs):
        start = time.monotonic()
        result = func(*args, **kwargs)
        logger.info("Computed in %s (HH:mm:ss)", datetime.timedelta(seconds=time.monotonic() - start))
        return result
    return timed
