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


# ---- WGS84 -> UTM without the `utm` wheel ---------------------------------------------------------------------------------------
# The reference reprojects a WGS84 grid with `utm.from_latlon(lat, lon)` (helpers.py:91-97; PyPI package `utm`, version not
# pinned by the reference, absent from this image).  What follows restates that function's published algorithm (Turbo87/utm,
# conversion.py: the Transverse Mercator series of USGS Professional Paper 1395 / Snyder on the WGS84 ellipsoid, scale 0.9996) for
# array input, including its choice of ONE zone for the whole array from the array's FIRST element (with the Norway / Svalbard
# exceptions) - that choice is what makes np.gradient of the eastings meaningful across a grid that straddles a zone boundary.
# Pinned by known answers of the package's own README and test-suite (tests/test_host_api.py); when the package is installed
# it is used instead.
_UTM_K0 = 0.9996
_UTM_E = 0.00669438
_UTM_E2 = _UTM_E * _UTM_E
_UTM_E3 = _UTM_E2 * _UTM_E
_UTM_E_P2 = _UTM_E / (1.0 - _UTM_E)
_UTM_M1 = 1 - _UTM_E / 4 - 3 * _UTM_E2 / 64 - 5 * _UTM_E3 / 256
_UTM_M2 = 3 * _UTM_E / 8 + 3 * _UTM_E2 / 32 + 45 * _UTM_E3 / 1024
_UTM_M3 = 15 * _UTM_E2 / 256 + 45 * _UTM_E3 / 1024
_UTM_M4 = 35 * _UTM_E3 / 3072
_UTM_R = 6378137.0


def _utm_zone_number(latitude, longitude):
    """Zone of the array's first element (the package's rule for numpy input), Norway and Svalbard as the package has them."""
    lat = float(np.asarray(latitude).flat[0])
    lon = float(np.asarray(longitude).flat[0])
    if 56 <= lat < 64 and 3 <= lon < 12:
        return 32
    if 72 <= lat <= 84 and lon >= 0:
        if lon < 9:
            return 31
        if lon < 21:
            return 33
        if lon < 33:
            return 35
        if lon < 42:
            return 37
    return int((lon + 180) / 6) + 1


def _utm_from_latlon(latitude, longitude):
    """``(easting, northing, zone_number)`` of WGS84 latitudes / longitudes in degrees (arrays of one shape), all in the zone of
    the first element.  Raises like the package outside 80 S ... 84 N / 180 W ... 180 E and for latitudes of mixed sign."""
    lat = np.asarray(latitude, dtype=np.float64)
    lon = np.asarray(longitude, dtype=np.float64)
    if lat.size == 0:
        raise ValueError("from_latlon: empty input")
    if not (np.min(lat) >= -80.0 and np.max(lat) <= 84.0):
        raise ValueError("latitude out of range (must be between 80 deg S and 84 deg N)")
    if not (np.min(lon) >= -180.0 and np.max(lon) <= 180.0):
        raise ValueError("longitude out of range (must be between 180 deg W and 180 deg E)")
    if np.min(lat) < 0 and np.max(lat) >= 0:
        raise ValueError("latitudes must all have the same sign")
    lat_rad = np.radians(lat)
    lat_sin, lat_cos = np.sin(lat_rad), np.cos(lat_rad)
    lat_tan = lat_sin / lat_cos
    lat_tan2 = lat_tan * lat_tan
    lat_tan4 = lat_tan2 * lat_tan2
    zone = _utm_zone_number(lat, lon)
    central_lon_rad = np.radians((zone - 1) * 6 - 180 + 3)
    n = _UTM_R / np.sqrt(1 - _UTM_E * lat_sin ** 2)
    c = _UTM_E_P2 * lat_cos ** 2
    d_lon = (np.radians(lon) - central_lon_rad + np.pi) % (2 * np.pi) - np.pi  # the package's mod_angle
    a = lat_cos * d_lon
    a2 = a * a
    a3 = a2 * a
    a4 = a3 * a
    a5 = a4 * a
    a6 = a5 * a
    m = _UTM_R * (_UTM_M1 * lat_rad - _UTM_M2 * np.sin(2 * lat_rad) + _UTM_M3 * np.sin(4 * lat_rad) - _UTM_M4 * np.sin(6 * lat_rad))
    easting = _UTM_K0 * n * (a + a3 / 6 * (1 - lat_tan2 + c) +
                             a5 / 120 * (5 - 18 * lat_tan2 + lat_tan4 + 72 * c - 58 * _UTM_E_P2)) + 500000
    northing = _UTM_K0 * (m + n * lat_tan * (a2 / 2 + a4 / 24 * (5 - lat_tan2 + 9 * c + 4 * c ** 2) +
                                             a6 / 720 * (61 - 58 * lat_tan2 + lat_tan4 + 600 * c - 330 * _UTM_E_P2)))
    if np.max(lat) < 0:
        northing = northing + 10000000
    return easting, northing, zone


def _wgs84_to_utm(x_coords, y_coords):
    lon, lat = np.meshgrid(x_coords, y_coords)
    try:
        import utm  # noqa: PLC0415
        east, north, _, _ = utm.from_latlon(lat, lon)
    except ImportError:
        east, north, _ = _utm_from_latlon(lat, lon)
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
