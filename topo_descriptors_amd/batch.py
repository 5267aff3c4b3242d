"""Batch wrappers: all scales of one descriptor with the DEM uploaded to the GPU once.

Counterparts of the reference's ``compute_dem / compute_tpi / compute_std / compute_gradient /
compute_sx`` (reference topo.py:16-59, 88-141, 216-269, 534-594, 715-772): same arguments, same
output names (``_tpi_name`` topo.py:184, ``_std_name`` :310, ``_gradient_names`` :647, ``_sx_name``
:956, ``_dem_name`` :83), NaN re-insertion through ``ind_nans`` (topo.py:139), one output per
scale.  The reference writes netCDF through xarray; here the writer is pluggable: with xarray
installed the same ``topo_<NAME>.nc`` files are produced, otherwise ``topo_<NAME>.npy``.  Every
wrapper also returns ``{name: array}``.
"""
import logging
import os

import numpy as np

from . import CFG, _lib, device as d, helpers as hlp, topo

logger = logging.getLogger(__name__)


# ---- output names (identical strings to the reference) ---------------------------------------
def _dem_name(scale):
    return f"DEM_{scale}M"


def _tpi_name(scale, smth_factor):
    add = f"_SMTHFACT{smth_factor:.3g}" if smth_factor else ""
    return f"TPI_{scale}M{add}"


def _std_name(scale, smth_factor):
    add = f"_SMTHFACT{smth_factor:.3g}" if smth_factor else ""
    return f"STD_{scale}M{add}"


def _gradient_names(scale, sig_ratio):
    tail = f"_{scale}M_SIGRATIO{sig_ratio:.3g}"
    return ["WE_DERIVATIVE" + tail, "SN_DERIVATIVE" + tail, "SLOPE" + tail, "ASPECT" + tail]


def _valley_ridge_names(scale, mode, smth_factor):
    """reference topo.py:450-458"""
    add = f"_SMTHFACT{smth_factor:.3g}" if smth_factor else ""
    return [f"{mode}_NORM_{scale}M{add}", f"{mode}_DIR_{scale}M{add}"]


def _sx_name(radius, azimuth):
    return f"SX_RADIUS{int(radius)}_AZIMUTH{int(azimuth)}"


# ---- writer ------------------------------------------------------------------------------------
def write_output(array, dem_ds, name, crop=None, outdir=".", units=None):
    """``hlp.to_netcdf`` of the reference (helpers.py:34-65) when xarray is there, .npy otherwise."""
    name = str.upper(name)
    if outdir is None:
        return None
    os.makedirs(outdir, exist_ok=True)
    if hlp._xr is not None and isinstance(dem_ds, hlp._xr.Dataset):  # pragma: no cover
        ds = hlp._xr.Dataset({name: (hlp.get_da(dem_ds).dims, array)}, coords=dem_ds.coords,
                             attrs=dem_ds.attrs).sel(crop)
        if units is not None:
            ds[name].attrs.update(units=units)
        path = os.path.join(outdir, f"topo_{name}.nc")
        ds.to_netcdf(path)
    else:
        if crop is not None:
            raise NotImplementedError("crop needs xarray coordinates")
        path = os.path.join(outdir, f"topo_{name}.npy")
        np.save(path, array)
    logger.info("saved: %s", path)
    return path


def _as_list(x, n=None):
    if not hasattr(x, "__iter__"):
        x = [x] if n is None else [x] * n
    return list(x)


class _ResidentDem:
    """The DEM on the GPU for the duration of one wrapper call."""

    def __init__(self, dem_val):
        self.host = _lib.as_f32(dem_val)
        self.dev = d.DeviceArray.from_host(self.host)
        self.block = d.Block(self.dev)
        self.shape = self.host.shape

    def plane(self):
        return d.DeviceArray(*self.shape)

    def close(self):
        self.dev.free()


def _finish(array, ind_nans, dem_ds, name, crop, outdir, units, results):
    if ind_nans is not None and len(ind_nans):
        array[ind_nans] = np.nan
    write_output(array, dem_ds, name, crop, outdir, units)
    results[name] = array


def compute_dem(dem_ds, scales, ind_nans=(), crop=None, outdir="."):
    """Gaussian-smoothed DEM per scale, sigma = scale_px / CFG.scale_std (reference topo.py:16-59)."""
    hlp.check_dem(dem_ds)
    scales = _as_list(scales)
    scales_pxl, _ = hlp.scale_to_pixel(scales, dem_ds)
    res = _ResidentDem(hlp.get_da(dem_ds).values)
    out, results = res.plane(), {}
    try:
        for scale, px in zip(scales, scales_pxl):
            sigma = px / CFG.scale_std
            res.block.gaussian(sigma, sigma, out)
            _finish(out.to_host(), ind_nans, dem_ds, _dem_name(scale), crop, outdir, "m", results)
    finally:
        out.free()
        res.close()
    return results


def _tpi_std(dem_ds, scales, smth_factors, ind_nans, crop, outdir, want):
    hlp.check_dem(dem_ds)
    scales = _as_list(scales)
    smth_factors = _as_list(smth_factors, len(scales))
    scales_pxl, _ = hlp.scale_to_pixel(scales, dem_ds)
    sigmas = hlp.get_sigmas(smth_factors, scales_pxl)
    res = _ResidentDem(hlp.get_da(dem_ds).values)
    out, smooth, results = res.plane(), None, {}
    try:
        # un-smoothed TPI at two small scales shares one pass over the DEM (Block.tpi_multi).  A pair is
        # computed when the loop reaches its first member: that plane is finished (NaN re-insertion, output) at once,
        # the partner's waits on the host for its turn - one extra host plane at most, and nothing computed is lost if a
        # later scale fails.
        partner, ahead, second = {}, {}, None
        if want == "tpi":
            small = {}
            for k, (px, sigma) in enumerate(zip(scales_pxl, sigmas)):
                if not sigma and int(px) in (5, 7, 9, 11):
                    small.setdefault(int(px), k)
            sizes = sorted(small)
            for a, b in zip(sizes[0::2], sizes[1::2]):
                ka, kb = sorted((small[a], small[b]))
                partner[ka] = kb
            if len(sizes) % 2:
                logger.info("TPI at %s px has no partner for a shared pass: computed alone", sizes[-1])
        for k, (scale, px, fact, sigma) in enumerate(zip(scales, scales_pxl, smth_factors, sigmas)):
            logger.info("Computing scale %s meters with smoothing factor %s ...", scale, fact)
            if k in ahead:
                _finish(ahead.pop(k), ind_nans, dem_ds, _tpi_name(scale, fact), crop, outdir, "m", results)
                continue
            if k in partner:
                kb = partner[k]
                second = second or res.plane()
                res.block.tpi_multi([int(px), int(scales_pxl[kb])], [out, second])
                ahead[kb] = second.to_host()
                _finish(out.to_host(), ind_nans, dem_ds, _tpi_name(scale, fact), crop, outdir, "m", results)
                continue
            block = res.block
            if sigma:  # pre-smoothing (reference topo.py:172-173, :297-298)
                smooth = smooth or res.plane()
                res.block.gaussian(sigma, sigma, smooth)
                block = d.Block(smooth)
            if want == "tpi":
                block.tpi_std(int(px), tpi=out)
                array, name = out.to_host(), _tpi_name(scale, fact)
            else:
                block.tpi_std(int(px), std=out)
                array, name = out.to_host().astype(np.float64), _std_name(scale, fact)
            _finish(array, ind_nans, dem_ds, name, crop, outdir, "m", results)
    finally:
        out.free()
        for plane in (smooth, second):
            if plane is not None:
                plane.free()
        res.close()
    return results


def compute_tpi(dem_ds, scales, smth_factors=None, ind_nans=(), crop=None, outdir="."):
    """TPI for every scale (reference topo.py:88-141)."""
    logger.info("***Starting TPI computation for scales %s meters***", scales)
    return _tpi_std(dem_ds, scales, smth_factors, ind_nans, crop, outdir, "tpi")


def compute_std(dem_ds, scales, smth_factors=None, ind_nans=(), crop=None, outdir="."):
    """Windowed standard deviation for every scale (reference topo.py:216-269)."""
    logger.info("***Starting STD computation for scales %s meters***", scales)
    return _tpi_std(dem_ds, scales, smth_factors, ind_nans, crop, outdir, "std")


def compute_gradient(dem_ds, scales, sig_ratios=1, ind_nans=(), crop=None, outdir="."):
    """dx, dy, slope, aspect for every scale (reference topo.py:534-594)."""
    hlp.check_dem(dem_ds)
    logger.info("***Starting gradients computation for scales %s meters***", scales)
    scales = _as_list(scales)
    sig_ratios = _as_list(sig_ratios, len(scales))
    scales_pxl, res_meters = hlp.scale_to_pixel(scales, dem_ds)
    sigmas = scales_pxl / CFG.scale_std
    dem_val = hlp.get_da(dem_ds).values
    results = {}
    two_d = np.ndim(res_meters["x"]) > 1 or np.ndim(res_meters["y"]) > 1
    res = None if two_d else _ResidentDem(dem_val)
    outs = None if two_d else [res.plane() for _ in range(4)]
    try:
        for scale, sigma, ratio in zip(scales, sigmas, sig_ratios):
            names = _gradient_names(scale, ratio)
            if two_d:  # per-pixel resolutions (WGS84 grids): host-buffer entry point
                arrays = topo.gradient(dem_val, sigma, res_meters, sig_ratio=ratio)
            else:
                res.block.gradient(sigma, res_meters["x"], res_meters["y"], sig_ratio=ratio,
                                   dx=outs[0], dy=outs[1], slope=outs[2], aspect=outs[3])
                arrays = [o.to_host() for o in outs]
            for array, name, units in zip(arrays, names, ["1", "1", "degree", "degree"]):
                _finish(array, ind_nans, dem_ds, name, crop, outdir, units, results)
    finally:
        if res is not None:
            for o in outs:
                o.free()
            res.close()
    return results


def compute_sx(dem_ds, azimuth, radius, height=10.0, azimuth_arc=10.0, azimuth_steps=15,
               radius_min=0.0, crop=None, outdir="."):
    """Sx for one azimuth (reference topo.py:715-772).  ``azimuth`` may also be a sequence: the
    planes of all azimuths then come from one pass over the DEM (``topo.sx_multi``), each saved
    under the name the reference gives it."""
    hlp.check_dem(dem_ds)
    logger.info("***Starting Sx computation for azimuth %s meters and radius %s***", azimuth, radius)
    results = {}
    if np.ndim(azimuth) == 0:
        array = topo.sx(dem_ds, azimuth, radius, height=height, azimuth_arc=azimuth_arc,
                        azimuth_steps=azimuth_steps, radius_min=radius_min)
        _finish(array, None, dem_ds, _sx_name(radius, azimuth), crop, outdir, "degree", results)
        return results
    arrays = topo.sx_multi(dem_ds, azimuth, radius, height=height, azimuth_arc=azimuth_arc,
                           azimuth_steps=azimuth_steps, radius_min=radius_min)
    for az, array in zip(azimuth, arrays):
        _finish(array, None, dem_ds, _sx_name(radius, az), crop, outdir, "degree", results)
    return results


def compute_valley_ridge(dem_ds, scales, mode, flat_list=[0, 0.15, 0.3], smth_factors=None, ind_nans=(),  # noqa: B006
                         crop=None, outdir="."):
    """Valley or ridge index (norm and direction) for every scale (reference topo.py:317-386).

    The DEM goes to the GPU once; per scale the optional pre-smoothing and the 180-angle pass run on the
    device-resident plane (the mean and standard deviation are numpy's, see below)."""
    hlp.check_dem(dem_ds)
    if mode not in ("valley", "ridge"):
        raise ValueError(f"Unknown mode {mode!r}")
    logger.info("***Starting %s index computation for scales %s meters***", mode, scales)
    scales = _as_list(scales)
    smth_factors = _as_list(smth_factors, len(scales))
    scales_pxl, _ = hlp.scale_to_pixel(scales, dem_ds)
    sigmas = hlp.get_sigmas(smth_factors, scales_pxl)
    res = _ResidentDem(hlp.get_da(dem_ds).values)
    norm, direction, smooth, results = res.plane(), res.plane(), None, {}
    angles = np.arange(0, 180, dtype=np.float32)

    def tables(px):
        kernels = topo._ridge_kernels(int(px), flat_list) if mode == "ridge" else topo._valley_kernels(int(px), flat_list)
        return (kernels.shape[0],) + topo._valley_ridge_tables(kernels, angles)

    # the rotated kernels of the next scale are built on the host while the GPU works on this one
    from concurrent.futures import ThreadPoolExecutor  # noqa: PLC0415
    pool = ThreadPoolExecutor(1)
    try:
        ahead = pool.submit(tables, scales_pxl[0]) if len(scales) else None
        for k, (scale, px, fact, sigma) in enumerate(zip(scales, scales_pxl, smth_factors, sigmas)):
            logger.info("Computing scale %s meters with smoothing factor %s ...", scale, fact)
            block, plane = res.block, res.dev
            if sigma:  # pre-smoothing (reference topo.py:424-425)
                smooth = smooth or res.plane()
                res.block.gaussian(sigma, sigma, smooth)
                block, plane = d.Block(smooth), smooth
            # numpy's own float32 mean / std of the whole (smoothed) array, like the reference (topo.py:427)
            # and like topo.valley_ridge: the wrapper and the single call then agree bit for bit, also where
            # two directions nearly tie
            field = plane.to_host() if sigma else res.host
            mean, stdev = float(field.mean()), float(field.std())
            n_planes, taps, ksize, ang = ahead.result()
            ahead = pool.submit(tables, scales_pxl[k + 1]) if k + 1 < len(scales) else None
            block.valley_ridge(taps, ksize, ang, n_planes, mean, stdev, norm, direction)
            for array, name in zip((norm.to_host(), direction.to_host()), _valley_ridge_names(scale, mode, fact)):
                _finish(array, ind_nans, dem_ds, name, crop, outdir, "1", results)
    finally:
        pool.shutdown(wait=True, cancel_futures=True)
        for a in (norm, direction, smooth):
            if a is not None:
                a.free()
        res.close()
    return results
