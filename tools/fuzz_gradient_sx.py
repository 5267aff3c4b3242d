"""Randomised parity sweep of topo.gradient / topo.dem / topo.sx / topo.valley_ridge against the
oracle (the reference's own scipy calls for the Gaussian chain, which rounds to float32 between
the axes like the GPU does), plus row-block bit-identity of the gradient (test-side tool).
    python tools/fuzz_gradient_sx.py [seconds=60] [seed=0]"""
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
fails, counts = [], {"gradient": 0, "gauss": 0, "sx": 0, "valley": 0}
t_end = time.time() + budget
REL = 1e-4


class Var:
    def __init__(self, values, dims):
        self.values, self.dims = values, dims


class Dataset:
    def __init__(self, dem, x, y):
        self._v = {"dem": Var(dem, ("y", "x")), "x": Var(x, ("x",)), "y": Var(y, ("y",))}
        self.attrs = {"crs": "epsg:2056"}

    def __getitem__(self, k):
        return self._v[k]

    def __iter__(self):
        return iter(["dem"])


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) / max(float(np.max(np.abs(b))), 1e-30)


while time.time() < t_end:
    ny = int(rng.choice([1, 2, 3, 5, 31, 61, 62, 63, 64, 65, 124, 127, 200, int(rng.integers(1, 320))]))
    nx = int(rng.choice([1, 2, 3, 5, 63, 125, 126, 127, 128, 129, 250, 256, 300, int(rng.integers(1, 420))]))
    dem = orc.synthetic_dem(ny, nx, seed=int(rng.integers(1 << 30)), integer=bool(rng.integers(2)),
                            row0=int(rng.integers(0, 3000)), col0=int(rng.integers(0, 3000)))
    what = rng.choice(["gradient", "gradient", "gauss", "sx", "valley"])
    try:
        if what == "gradient":
            sigma = float(rng.choice([0.5, 0.75, 1.0, 1.25, 2.25, 3.25, 4.0, 5.75, 7.5, 12.0, 13.0, 20.0, 30.25]))  # (from 12.25: split-once axis 1)
            ratio = float(rng.choice([1, 1, 1, 0.5, 2]))
            dx0 = float(rng.choice([30.0, 25.0, 50.0]))
            dy0 = float(rng.choice([-30.0, -25.0, 30.0]))
            mode = rng.choice(["scalar", "1d"])
            res = {"x": np.float64(dx0), "y": np.float64(dy0)} if mode == "scalar" else \
                {"x": np.full(nx, dx0), "y": np.full(ny, dy0)}
            ctx = f"gradient ny={ny} nx={nx} sigma={sigma} ratio={ratio} res={mode}({dx0},{dy0})"
            if sigma > 1 and (ny < 2 or nx < 2):
                continue  # numpy.gradient raises in the reference too (the product raises TopoAmdError)
            counts["gradient"] += 1
            # float32 rounding of the smoothed field (the reference rounds it too), seen through a
            # central difference: the noise floor of dx, dy whatever their size
            noise = 4.0 * np.finfo(np.float32).eps * float(np.max(np.abs(dem))) / min(abs(dx0), abs(dy0))
            got = topo.gradient(dem, sigma, res, ratio)
            want = orc.gradient_scipy(dem, sigma, res, ratio)
            for k, nm in enumerate(("dx", "dy", "slope")):
                w = np.asarray(want[k], np.float64)
                tol = REL * float(np.max(np.abs(w))) + (noise if nm != "slope" else np.degrees(noise))
                err = float(np.max(np.abs(got[k] - w)))
                if not err <= tol:
                    fails.append(f"{ctx} {nm}: err {err:.3g} tol {tol:.3g}")
            steep = np.asarray(want[2]) > 0.1
            if steep.any():
                a = float(np.max(orc.wrapped_angle_diff(got[3], want[3])[steep]))
                # aspect is ill-conditioned at small slopes: d(aspect) ~ d(gradient) / |gradient|, so
                # aspect error x slope (both in degrees) ~ (180/pi)^2 x the error of dx, dy
                cond = float(np.max(orc.wrapped_angle_diff(got[3], want[3])[steep] * np.asarray(want[2])[steep]))
                g_tol = REL * max(float(np.max(np.abs(want[0]))), float(np.max(np.abs(want[1])))) + noise
                if not (a <= REL * 360.0 or cond <= 2.0 * (180.0 / np.pi) ** 2 * g_tol):
                    fails.append(f"{ctx} aspect: {a:.3g} deg (x slope {cond:.3g})")
            if np.any((got[3] < 0) | (got[3] >= 360)):
                fails.append(f"{ctx} aspect outside [0, 360)")
            # row blocks, bit for bit, through the device API
            if ny >= 2:
                nb = int(rng.integers(2, min(ny, 4) + 1))
                up, down = shard.halo_rows(_lib.DESC_GRADIENT, sigma * max(ratio, 1.0) if sigma > 1 else sigma)
                up, down = shard.halo_rows(_lib.DESC_GRADIENT, max(sigma, sigma * ratio))
                rx = np.broadcast_to(np.asarray(res["x"], np.float64), (nx,)).copy()
                ry = np.broadcast_to(np.asarray(res["y"], np.float64), (ny,)).copy()
                parts = [[], [], [], []]
                for row0, rows in shard.split_rows(ny, nb):
                    lo, hi = max(0, row0 - up), min(ny, row0 + rows + down)
                    dev = d.DeviceArray.from_host(dem[lo:hi])
                    blk = d.Block(dev, row0=lo, gny=ny)
                    o = [d.DeviceArray(rows, nx) for _ in range(4)]
                    blk.gradient(sigma, rx, ry, ratio, dx=o[0], dy=o[1], slope=o[2], aspect=o[3], out_row0=row0, out_rows=rows)
                    d.sync()
                    for k in range(4):
                        parts[k].append(o[k].to_host())
                        o[k].free()
                    dev.free()
                for k, nm in enumerate(("dx", "dy", "slope", "aspect")):
                    if not np.array_equal(np.concatenate(parts[k]), got[k]):
                        fails.append(f"{ctx} row blocks nb={nb} {nm}")
        elif what == "gauss":
            sigma = float(rng.choice([0.75, 1.5, 2.25, 3.25, 6.0, 9.5, 15.0, 20.0, 30.25]))
            ctx = f"gauss ny={ny} nx={nx} sigma={sigma}"
            counts["gauss"] += 1
            from scipy import ndimage
            got = topo.dem(dem, sigma)
            want = ndimage.gaussian_filter(dem, sigma)
            if not rel(got, want) <= REL:
                fails.append(f"{ctx}: rel {rel(got, want):.3g}")
        elif what == "sx":
            az = float(rng.choice([0, 45, 90, 135, 180, 225.7, 270, 315, float(rng.uniform(0, 360))]))
            radius = float(rng.choice([60, 150, 300, 500]))
            south_up = bool(rng.integers(2))
            x = 2600000.0 + 30.0 * np.arange(nx)
            y = 1200000.0 + (30.0 if south_up else -30.0) * np.arange(ny)
            if nx < 2 or ny < 2:
                continue  # grid spacing undefined
            ctx = f"sx ny={ny} nx={nx} az={az:.1f} radius={radius} south_up={south_up}"
            counts["sx"] += 1
            got = topo.sx(Dataset(dem, x, y), az, radius)
            want = orc.sx(dem, x, y, az, radius)
            tol = REL * max(float(np.max(np.abs(want))), 1.0)
            if not float(np.max(np.abs(got - want))) <= tol:
                fails.append(f"{ctx}: err {float(np.max(np.abs(got - want))):.3g} tol {tol:.3g}")
        else:
            if ny < 3 or nx < 3 or ny * nx > 40000:
                continue
            size = int(rng.choice([3, 5, 7, 9, 11, 13, 15, 17, 19, 25, 33]))   # (from 19 px on: the streamed form)
            flats = [[0, 0.15, 0.3], [0], [0.2, 0.4], [0, 0.1, 0.2, 0.3]][int(rng.integers(4))]
            mode = str(rng.choice(["valley", "ridge"]))
            angles = np.sort(rng.choice(np.arange(180, dtype=np.float32), int(rng.integers(1, 40)), replace=False))
            # the matrix-pipe kernel (what these sizes take) three times out of four, the tap-by-tap kernel otherwise; read at every launch
            # and of the matrix-pipe calls one in three over the cells instead of the pairs (TOPO_AMD_VALLEY_FOLD=0)
            os.environ["TOPO_AMD_VALLEY_MFMA_MAX_KERNEL"] = "0" if rng.random() < 0.25 else "64"
            os.environ["TOPO_AMD_VALLEY_FOLD"] = "0" if rng.random() < 0.33 else "1"
            ctx = (f"valley ny={ny} nx={nx} size={size} flats={flats} mode={mode} angles={len(angles)} "
                   f"matrix={os.environ['TOPO_AMD_VALLEY_MFMA_MAX_KERNEL']} fold={os.environ['TOPO_AMD_VALLEY_FOLD']}")
            counts["valley"] += 1
            kernels = topo._ridge_kernels(size, flats) if mode == "ridge" else topo._valley_kernels(size, flats)
            taps, ksize, ang = topo._valley_ridge_tables(kernels, angles)
            dev = d.DeviceArray.from_host(dem)
            n_out, a_out = d.DeviceArray(ny, nx), d.DeviceArray(ny, nx)
            mean, stdev = float(dem.mean()), float(dem.std())
            d.Block(dev).valley_ridge(taps, ksize, ang, len(flats), mean, stdev, n_out, a_out)
            d.sync()
            norm, direction = n_out.to_host(), a_out.to_host()
            for x_ in (dev, n_out, a_out):
                x_.free()
            (norm_ex, _), maps = orc.valley_ridge_exact(dem, size, mode, flats, angles=angles, return_maps=True)
            scale = max(float(np.max(norm_ex)), 1e-6)
            if not float(np.max(np.abs(norm - norm_ex))) <= REL * scale + 1e-5:
                fails.append(f"{ctx}: norm err {float(np.max(np.abs(norm - norm_ex))):.3g} scale {scale:.3g}")
            idx = np.searchsorted(angles, direction)
            ok = (idx < len(angles)) & (angles[np.minimum(idx, len(angles) - 1)] == direction)
            if not ok.all():
                fails.append(f"{ctx}: direction not one of the angles")
            else:
                gap = float(np.max(np.max(maps, axis=0) - np.take_along_axis(maps, idx[None], axis=0)[0]))
                if not gap <= REL * scale + 1e-5:
                    fails.append(f"{ctx}: direction not a maximiser (gap {gap:.3g})")
    except Exception as exc:  # noqa: BLE001
        fails.append(f"exception {what} ny={ny} nx={nx}: {type(exc).__name__}: {str(exc)[:200]}")

import collections  # noqa: E402
print(dict(collections.Counter(f.split()[0] + (" " + f.split()[1] if f.startswith("exception") else "") for f in fails)))
for f in fails[:40]:
    print("FAIL", f)
print(f"{counts} cases, {len(fails)} failures")
sys.exit(1 if fails else 0)
