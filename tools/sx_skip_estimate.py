"""How often could the Sx kernel skip a chain of ray pixels?  (task 6: a wave-level early-out.)

A wave of sx_kernel owns 16 rows x 64 columns of pixels and walks the chains of 8 neighbouring ray pixels of the
sector; a chain can be skipped for the wave when NONE of its 1024 pixels can still gain from it, i.e. when
    (max z over the samples the chain reads for this wave - z0 - height) / (smallest distance of the chain) <= best(z0)
for every pixel, `best` being the maximum over the chains walked so far (near to far).  This script evaluates that on the
CPU for the bench DEM (whole metres) and for a rugged one (10 x the relief), azimuth 0, radius 2000 m at 30 m, with
three bounds: "exact" (skip iff no pixel gains from the chain: the ceiling of any scheme), "region" (maximum over
exactly the samples the chain reads for the wave) and "rows" (maxima of 8-row blocks over the whole staged tile width,
what the kernel could keep in LDS cheaply).  Prints the share of (wave, chain) pairs skipped.
    python tools/sx_skip_estimate.py [waves=200]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import topo_oracle as orc  # noqa: E402

NW = int(sys.argv[1]) if len(sys.argv) > 1 else 200
HEIGHT = 10.0


def chains(offs, dist):
    """Chains of up to 8 ray pixels that are neighbours down a column (the kernel's axis for azimuth 0), near to far."""
    pts = {}
    for (dj, di), dd in zip(offs, dist):
        if not np.isnan(dd):
            pts[(int(dj), int(di))] = float(dd)
    out = []
    for di in sorted({k[1] for k in pts}):
        col = sorted(k[0] for k in pts if k[1] == di)
        run = [col[0]]
        for dj in col[1:] + [None]:
            if dj is not None and dj == run[-1] + 1 and len(run) < 8:
                run.append(dj)
            else:
                out.append((di, run, [pts[(j, di)] for j in run]))
                run = [dj]
    out.sort(key=lambda c: min(c[2]))
    return out


def study(dem, name, rng):
    window, offs, dist = orc.sx_geometry(0.0, 2000.0, 30.0, -30.0)
    ch = chains(offs, dist)
    ny, nx = dem.shape
    tot = skip_exact = skip_region = skip_rows = 0
    for _ in range(NW):
        y0 = int(rng.integers(window, ny - window - 16))
        x0 = int(rng.integers(window, nx - window - 64))
        z0 = dem[y0:y0 + 16, x0:x0 + 64].astype(np.float64)
        best = np.full(z0.shape, -np.inf)
        for di, run, dd in ch:
            gain = np.full(z0.shape, -np.inf)
            for dj, d1 in zip(run, dd):
                zp = dem[y0 + dj:y0 + dj + 16, x0 + di:x0 + di + 64]
                gain = np.maximum(gain, (zp - z0 - HEIGHT) / d1)
            dmin = min(dd)
            region = dem[y0 + run[0]:y0 + run[-1] + 16, x0 + di:x0 + di + 64].max()
            r0 = (y0 + run[0]) // 8 * 8
            r1 = -(-(y0 + run[-1] + 16) // 8) * 8
            rows = dem[r0:r1, max(0, x0 - 1):x0 + 65].max()   # (azimuth 0: di in [-1, 1])
            tot += 1
            skip_exact += bool(np.all(gain <= best))
            # a bound can only be used on its sign-safe side: (M - z0 - h) / dmin bounds the gain when M - z0 - h > 0;
            # when it is negative the true gain is even smaller in magnitude only if divided by the LARGEST distance
            def bound(m):
                num = m - z0 - HEIGHT
                return np.where(num > 0, num / dmin, num / max(dd))
            skip_region += bool(np.all(bound(region) <= best))
            skip_rows += bool(np.all(bound(rows) <= best))
            best = np.maximum(best, gain)
    print(f"{name}: {len(ch)} chains per wave, {tot} (wave, chain) pairs: skipped exact {skip_exact / tot:.3f}, "
          f"region bound {skip_region / tot:.3f}, 8-row-block bound {skip_rows / tot:.3f}")


rng = np.random.default_rng(0)
n = 1536
dem = orc.synthetic_dem(n, n, seed=0)
study(dem, "bench DEM (relief 1 x)", rng)
mean = dem.mean()
study((mean + 10.0 * (dem - mean)).astype(np.float32), "rugged DEM (relief 10 x)", rng)
