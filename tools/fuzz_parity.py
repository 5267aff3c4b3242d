"""Randomised parity sweep of topo.tpi / topo.std / topo.tpi_std against the exact oracle, plus
row-block bit-identity, over shapes, disc sizes and value classes the structured tests do not
enumerate (test-side tool: imports the oracle).  Run on the GPU box:
    python tools/fuzz_parity.py [seconds=60] [seed=0] [big]
Prints one line per failure and a summary; exit code 1 if anything failed."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import topo_oracle as orc  # noqa: E402
from topo_descriptors_amd import _lib, device as d, shard, topo  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails, cases = [], 0
t_end = time.time() + budget


def make_dem(ny, nx):
    kind = rng.choice(["int", "frac", "mixed_cols", "mixed_rows", "negative", "sea", "steps"])
    z = orc.synthetic_dem(ny, nx, seed=int(rng.integers(1 << 30)), integer=True, row0=int(rng.integers(0, 5000)),
                          col0=int(rng.integers(0, 5000))).astype(np.float32)
    if kind == "frac":
        z = z + rng.random((ny, nx)).astype(np.float32)
    elif kind == "mixed_cols":
        c = int(rng.integers(0, nx + 1))
        z[:, :c] += np.float32(rng.random())
    elif kind == "mixed_rows":
        a, b = sorted(int(v) for v in rng.integers(0, ny + 1, 2))
        z[a:b] += np.float32(rng.random())
    elif kind == "negative":
        z = -z + np.float32(rng.choice([0.0, 0.25, 0.37]))
    elif kind == "sea":
        z = np.maximum(z - 1900.0, 0.0).astype(np.float32)  # large flat zero areas
    elif kind == "steps":
        z = (np.floor(z / 250.0) * 250.0).astype(np.float32) + np.float32(rng.choice([0.0, 0.5]))
    extra = rng.choice(["none", "none", "nodata", "nan"])
    if extra == "nodata" and ny > 4 and nx > 4:
        j, i = int(rng.integers(0, ny - 2)), int(rng.integers(0, nx - 2))
        z[j:j + int(rng.integers(1, ny - j)), i:i + int(rng.integers(1, nx - i))] = -9999.0
    if extra == "nan":
        z[int(rng.integers(0, ny)), int(rng.integers(0, nx))] = np.nan
    return np.ascontiguousarray(z, dtype=np.float32), f"{kind}+{extra}"


extra_nan = 0


def check(name, got, want, tol, ctx):
    """Finite on both sides: within tol.  NaN in the oracle: NaN in the product.  The product may be
    NaN on more pixels (a NaN travels down the column prefix sums of its tile's fall-back path; the
    reference's FFT makes the whole array NaN), which is counted, not failed."""
    global extra_nan
    got = got.astype(np.float64)
    missing = np.isnan(want) & ~np.isnan(got)
    if missing.any():
        fails.append(f"{name} {ctx}: {int(missing.sum())} pixels finite where the oracle is NaN")
    extra_nan += int((np.isnan(got) & ~np.isnan(want)).sum())
    both = ~np.isnan(got) & ~np.isnan(want)
    err = np.abs(got - want)[both]
    if err.size and not np.all(err <= tol):
        fails.append(f"{name} {ctx}: max err {np.max(err):.3g} tol {tol:.3g}")


# third argument "big": the sizes around and beyond the wave-shift range (LDS-gather kernel up to 68,
# prefix planes from 70 / beyond 101) instead of the common ones
SIZES = ([66, 68, 70, 84, 100, 101, 103, 121, 151, 200, 255] if len(sys.argv) > 3 and sys.argv[3] == "big"
         else [1, 2, 3, 4, 5, 6, 7, 9, 11, 15, 17, 19, 25, 31, 33, 41, 67])
if os.environ.get("FUZZ_SIZES"):  # (lab: a comma-separated list)
    SIZES = [int(v) for v in os.environ["FUZZ_SIZES"].split(",")]
while time.time() < t_end:
    ny = int(rng.choice([1, 2, 3, 7, 33, 59, 60, 61, 120, 127, 200, 333, int(rng.integers(1, 400))]))
    nx = int(rng.choice([1, 3, 4, 5, 64, 189, 190, 191, 192, 250, 256, 380, int(rng.integers(1, 500))]))
    size = int(rng.choice(SIZES))
    dem, kind = make_dem(ny, nx)
    ctx = f"ny={ny} nx={nx} size={size} dem={kind}"
    cases += 1
    try:
        t_only = topo.tpi(dem, size)
        s_only = topo.std(dem, size)
        t_f, s_f = topo.tpi_std(dem, size)
    except Exception as exc:  # noqa: BLE001
        fails.append(f"exception {ctx}: {type(exc).__name__}: {exc}")
        continue
    if size == 1:
        continue  # division by zero in the reference (non-finite everywhere); nothing to compare
    has_bad = bool(np.isnan(dem).any() or (np.abs(dem) > 5000).any())
    want_t = orc.tpi_exact(dem, size)
    want_s = orc.std_exact(dem, size)
    scale_t = max(float(np.nanmax(np.abs(want_t))) if np.isfinite(want_t).any() else 1.0, 1.0)
    scale_s = max(float(np.nanmax(np.abs(want_s))) if np.isfinite(want_s).any() else 1.0, 1.0)
    # exact paths: float32 output rounding; nodata / NaN tiles may run the float fall-back chains
    tol_t = (2e-3 if has_bad else 2.5e-7) * scale_t + 2.5e-4
    # STD next to -9999 nodata: the float32 fall-back chains of the generic kernel (nx % 4 != 0) reach
    # ~6e-3 of the (then ~6000 m) STD; a known limit of the absurd-sample path
    tol_s = (1e-2 if has_bad else 2.5e-7) * scale_s + 2.5e-4
    # TPI alone from 19 px on: tiles with fractional elevations sum x in units of 2^-8 m (tpi_scaled_march_kernel)
    scaled = size % 2 == 1 and 19 <= size <= 101 and bool((dem != np.trunc(dem))[np.isfinite(dem)].any()) and \
        os.environ.get("TOPO_AMD_TPI_FRACTION_EXACT", "0") == "0"
    check("tpi", t_only, want_t, tol_t + (2.0 ** -9 if scaled else 0.0), ctx)
    check("std", s_only.astype(np.float64), want_s, tol_s, ctx)
    check("tpi(fused)", t_f, want_t, tol_t, ctx)
    if not has_bad:
        if scaled:
            if np.nanmax(np.abs(t_only - t_f)) > 2.0 ** -9:
                fails.append(f"|tpi - tpi_std()[0]| > 2^-9 {ctx}: max {np.nanmax(np.abs(t_only - t_f)):.3g}")
        elif not np.array_equal(t_only, t_f, equal_nan=True):
            fails.append(f"tpi != tpi_std()[0] {ctx}: {int((t_only != t_f).sum())} px, max {np.nanmax(np.abs(t_only - t_f)):.3g}")
        if not np.array_equal(s_only, s_f.astype(s_only.dtype), equal_nan=True):
            fails.append(f"std != tpi_std()[1] {ctx}")
    # row blocks: bit-identical to the single block (TPI alone and the fused pair)
    if ny >= 2 and not has_bad:
        nb = int(rng.integers(2, min(ny, 5) + 1))
        up, down = shard.halo_rows(_lib.DESC_TPI, size)
        tp, sp, tq = [], [], []
        # what an application that holds a raster in pieces does once: the class of the WHOLE raster, added up from the
        # rows each block owns (round 5: the unit of the scaled TPI route follows the raster's value range - a window of a
        # few hundred metres of relief takes 2^-13 m where an undeclared block would assume an ordinary DEM's 2^-8)
        scan = d.RasterScan()
        for row0, rows in shard.split_rows(ny, nb):
            dev = d.DeviceArray.from_host(dem[row0:row0 + rows])
            scan.add(d.Block(dev, row0=row0, gny=ny))
            dev.free()
        for row0, rows in shard.split_rows(ny, nb):
            lo, hi = max(0, row0 - up), min(ny, row0 + rows + down)
            dev = d.DeviceArray.from_host(dem[lo:hi])
            blk = d.Block(dev, row0=lo, gny=ny)
            scan.declare(blk)  # for this block's memory; it goes with dev.free()
            a, b, c = d.DeviceArray(rows, nx), d.DeviceArray(rows, nx), d.DeviceArray(rows, nx)
            blk.tpi_std(size, tpi=a, std=b, out_row0=row0, out_rows=rows)
            blk.tpi_std(size, tpi=c, out_row0=row0, out_rows=rows)
            d.sync()
            tp.append(a.to_host()); sp.append(b.to_host()); tq.append(c.to_host())
            for x in (a, b, c, dev):
                x.free()
        if not np.array_equal(np.concatenate(tp), t_f, equal_nan=True):
            fails.append(f"row blocks tpi(fused) {ctx} nb={nb}")
        if not np.array_equal(np.concatenate(sp).astype(np.float64), s_f.astype(np.float64), equal_nan=True):
            fails.append(f"row blocks std {ctx} nb={nb}")
        if not np.array_equal(np.concatenate(tq), t_only, equal_nan=True):
            fails.append(f"row blocks tpi {ctx} nb={nb}")

for f in fails[:40]:
    print("FAIL", f)
print(f"{cases} cases, {len(fails)} failures; {extra_nan} pixels NaN in the product only (NaN inputs)")
sys.exit(1 if fails else 0)
