#!/usr/bin/env python3
"""Row blocks against one block on DEMs that exercise every data-dependent kernel route: nodata strips next to terrain
(more relief than the integer chains hold), NaN / inf samples, finite samples beyond 2^18, fractional elevations, a raster
in millimetres - for TPI, TPI + STD, the Gaussian and the gradient.  Prints one line per case: the number of pixels whose
bits differ between the cut DEM and the single block (0 everywhere = routing is a function of the data and the global
grid only).

    python tools/probe_cut_invariance.py [rows=400] [nx=512]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import topo_oracle as orc  # noqa: E402
from topo_descriptors_amd import _lib, device as d, shard  # noqa: E402


def run_blocks(dem, nblocks, above, below, call):
    gny, nx = dem.shape
    pieces = None
    scan = None
    if nblocks > 1:  # what an application holding the raster in pieces does once: the class of the whole raster
        scan = d.RasterScan()
        for row0, rows in shard.split_rows(gny, nblocks):
            dev = d.DeviceArray.from_host(dem[row0:row0 + rows])
            scan.add(d.Block(dev, row0=row0, gny=gny))
            dev.free()
    for row0, rows in shard.split_rows(gny, nblocks):
        lo, hi = max(0, row0 - above), min(gny, row0 + rows + below)
        dev = d.DeviceArray.from_host(dem[lo:hi])
        blk = d.Block(dev, row0=lo, gny=gny)
        if scan is not None:
            scan.declare(blk)  # (keyed by the block's memory: gone with dev.free())
        outs = call(blk, row0, rows)
        d.sync()
        host = [o.to_host() for o in outs]
        pieces = [[h] for h in host] if pieces is None else [p + [h] for p, h in zip(pieces, host)]
        for o in outs:
            o.free()
        dev.free()
    return [np.concatenate(p, axis=0) for p in pieces]


def dems(gny, nx):
    out = {}
    base_i = orc.synthetic_dem(gny, nx, seed=5, integer=True)
    base_f = orc.synthetic_dem(gny, nx, seed=6, integer=False)
    a = base_i.copy(); a[gny // 2 + 3, nx // 3] = np.nan
    out["int+nan"] = a
    a = base_f.copy(); a[gny // 2 + 3, nx // 3] = np.nan
    out["frac+nan"] = a
    a = base_i.copy(); a[:, : nx // 8] = -9999.0
    out["int+nodata_cols"] = a
    a = base_i.copy(); a[gny // 2 - 20: gny // 2 + 9, :] = -9999.0
    out["int+nodata_rows_at_seam"] = a
    a = base_f.copy(); a[gny // 2 - 20: gny // 2 + 9, nx // 4:] = -9999.0
    out["frac+nodata_rows_at_seam"] = a
    a = base_f.copy(); a[gny // 3 + 7, 40:90] = -9999.0
    out["frac+nodata_line"] = a
    a = base_i.copy(); a[gny // 2 + 5, nx // 2] = 1.0e20
    out["int+1e20"] = a
    a = base_i.copy(); a[gny // 3: gny // 3 + 30, nx // 2:] = -3.4028235e38
    out["int+fltmin_block"] = a
    a = base_i.copy(); a[gny // 2 + 1, nx // 2 + 5] = np.inf
    out["int+inf"] = a
    a = base_f.copy(); a[: gny // 2 + 11] = np.rint(a[: gny // 2 + 11])
    out["half_int_half_frac"] = a
    out["mm"] = (base_f * 1000.0).astype(np.float32)
    a = base_f.copy(); a[gny // 2 + 9:] *= 1000.0
    out["half_m_half_mm"] = a.astype(np.float32)
    return out


def main():
    gny = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    nx = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    bad_total = 0
    for name, dem in dems(gny, nx).items():
        for size in (6, 7, 17, 31, 67):
            up, down = shard.halo_rows(_lib.DESC_TPI, size)
            for what in ("tpi", "tpi_std", "std"):
                def call(blk, row0, rows, what=what, size=size):
                    t = d.DeviceArray(rows, nx) if what != "std" else None
                    s = d.DeviceArray(rows, nx) if what != "tpi" else None
                    blk.tpi_std(size, tpi=t, std=s, out_row0=row0, out_rows=rows)
                    return [p for p in (t, s) if p is not None]
                whole = run_blocks(dem, 1, up, down, call)
                for nb in (2, 3, 5):
                    parts = run_blocks(dem, nb, up, down, call)
                    for k, (p, w) in enumerate(zip(parts, whole)):
                        diff = ~((p == w) | (np.isnan(p) & np.isnan(w)))
                        n = int(diff.sum())
                        if n:
                            bad_total += n
                            rows_bad = np.flatnonzero(diff.any(axis=1))
                            worst = float(np.nanmax(np.abs(np.where(diff, p - w, 0.0)))) if np.isfinite(np.where(diff, p - w, 0.0)).any() else float("nan")
                            nan_mismatch = int((np.isnan(p) != np.isnan(w)).sum())
                            print(f"DIFF {name:26s} size {size:3d} {what:8s} plane {k} blocks {nb}: {n:7d} px, rows {rows_bad[0]}..{rows_bad[-1]}, "
                                  f"max|d| {worst:.3g}, nan-mismatch {nan_mismatch}", flush=True)
        for sigma in (3.25, 13.0):
            up, down = shard.halo_rows(_lib.DESC_GRADIENT, sigma)
            def callg(blk, row0, rows, sigma=sigma):
                outs = [d.DeviceArray(rows, nx) for _ in range(4)]
                blk.gradient(sigma, [30.0], [-30.0], dx=outs[0], dy=outs[1], slope=outs[2], aspect=outs[3], out_row0=row0, out_rows=rows)
                return outs
            whole = run_blocks(dem, 1, up, down, callg)
            for nb in (2, 3, 5):
                parts = run_blocks(dem, nb, up, down, callg)
                for k, (p, w) in enumerate(zip(parts, whole)):
                    diff = ~((p == w) | (np.isnan(p) & np.isnan(w)))
                    n = int(diff.sum())
                    if n:
                        bad_total += n
                        rows_bad = np.flatnonzero(diff.any(axis=1))
                        print(f"DIFF {name:26s} gradient sigma {sigma} plane {k} blocks {nb}: {n} px, rows {rows_bad[0]}..{rows_bad[-1]}", flush=True)
        print(f"done {name}", flush=True)
    print("TOTAL differing pixels:", bad_total)


if __name__ == "__main__":
    main()
