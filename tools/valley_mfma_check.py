"""valley / ridge: the matrix-pipe kernel (csrc/valley_mfma.hip) against the direct kernel (csrc/valley.hip) and the float64 oracle,
and their times.  TOPO_AMD_VALLEY_MFMA_MAX_KERNEL is read at every launch (0 = direct kernel only).
    python tools/valley_mfma_check.py [size of the timed DEM, default 8192]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import topo_oracle as orc  # noqa: E402
from topo_descriptors_amd import _lib, device as d, topo  # noqa: E402


def run(dem, size, flats, route, angles=None, mode="valley", moments=None):
    os.environ["TOPO_AMD_VALLEY_MFMA_MAX_KERNEL"] = "0" if route == "direct" else "64"
    os.environ["TOPO_AMD_VALLEY_FOLD"] = os.environ.get("VM_FOLD", "1")   # the parity sweep: the folded form unless VM_FOLD=0
    kernels = topo._valley_kernels(size, flats) if mode == "valley" else topo._ridge_kernels(size, flats)
    ang = np.arange(0, 180, dtype=np.float32) if angles is None else angles
    taps, ksize, ang = topo._valley_ridge_tables(kernels, ang)
    dev = d.DeviceArray.from_host(dem)
    n, a = d.DeviceArray(*dem.shape), d.DeviceArray(*dem.shape)
    mean, stdev = moments if moments else (float(dem.mean()), float(dem.std()))
    d.Block(dev).valley_ridge(taps, ksize, ang, len(flats), mean, stdev, n, a)
    d.sync()
    out = n.to_host(), a.to_host()
    for x in (n, a, dev):
        x.free()
    return out, int(ksize.max())


def main():
    rng = np.random.default_rng(3)
    bad = 0
    if os.environ.get("VM_TIME_ONLY"):
        return timing()
    for size, flats in [(3, [0, 0.15, 0.3]), (5, [0, 0.15, 0.3]), (7, [0, 0.15, 0.3]), (7, [0]), (7, [0, 0.2]), (7, [0, 0.1, 0.2, 0.3]),
                        (9, [0, 0.15, 0.3]), (11, [0, 0.15, 0.3]), (13, [0, 0.15, 0.3]), (15, [0, 0.15, 0.3]), (17, [0, 0.15, 0.3])]:
        for kind in ("int", "frac", "nan"):
            dem = orc.synthetic_dem(150, 210, seed=size)
            if kind != "int":
                dem = (dem + rng.uniform(0, 1, dem.shape)).astype(np.float32)
            moments = float(dem.mean()), float(dem.std())
            if kind == "nan":
                dem[40, 50] = np.nan
                dem[100:103, 150] = np.inf
                dem[0, 0] = np.nan
                dem[149, 209] = -np.inf
                dem[70, 100] = 3e9
            for angles in (None, np.arange(0, 180, 7, dtype=np.float32)):
                (n0, a0), kmax = run(dem, size, flats, "direct", angles, moments=moments)
                (n1, a1), _ = run(dem, size, flats, "mfma", angles, moments=moments)
                fin = np.isfinite(n0)
                same_nan = np.array_equal(np.isnan(n0), np.isnan(n1)) and np.array_equal(np.isnan(a0), np.isnan(a1))
                scale = float(np.max(n0[fin]))
                err = float(np.max(np.abs(n0[fin] - n1[fin])))
                agree = float(np.mean(a0[fin] == a1[fin]))
                ok = same_nan and err <= 2e-5 * scale and agree > (0.99 if size == 3 else 0.995)
                note = ""
                if kind == "frac" and angles is not None and size <= 9:   # the float64 oracle, through its per-angle maps
                    ang = angles
                    (nx_, _), maps = orc.valley_ridge_exact(dem, size, "valley", flats, angles=ang, return_maps=True)
                    e0, e1 = float(np.max(np.abs(n0 - nx_))), float(np.max(np.abs(n1 - nx_)))
                    idx = np.searchsorted(ang, a1).astype(int)
                    at = np.take_along_axis(maps, idx[None], axis=0)[0]
                    miss = float(np.max(np.max(maps, axis=0) - at))
                    note = f" | vs float64: direct {e0:.2e}, matrix pipe {e1:.2e}, response lost by its direction {miss:.2e}"
                    ok = ok and e1 <= 1e-4 * scale and miss <= 1e-4 * scale
                bad += not ok
                print(f"size {size:2d} canvas {kmax:2d} planes {len(flats)} {kind:4s} angles {180 if angles is None else len(angles):3d}: "
                      f"max|norm diff| {err:.2e} of {scale:.2f}, same direction {agree:.4f}, nan pattern {'same' if same_nan else 'DIFFERENT'}"
                      f"{note}{'' if ok else '   <-- FAIL'}", flush=True)
    print("failures:", bad)
    timing()


def timing():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    if os.environ.get("VM_TIME_ONLY"):
        dem = orc.synthetic_dem(side, side, seed=1)
        kernels = topo._valley_kernels(7, [0, 0.15, 0.3])
        taps, ksize, ang = topo._valley_ridge_tables(kernels, np.arange(0, 180, dtype=np.float32))
        dev = d.DeviceArray.from_host(dem)
        n, a = d.DeviceArray(side, side), d.DeviceArray(side, side)
        blk = d.Block(dev)
        for lab in os.environ["VM_TIME_ONLY"].split(","):   # (one line per entry: several builds in one session via TOPO_AMD_LIBRARY)
            ts = []
            for _ in range(4):
                d.sync()
                t0 = time.perf_counter()
                blk.valley_ridge(taps, ksize, ang, 3, 1500.0, 400.0, n, a)
                d.sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            print(f"lab {lab}: {min(ts[1:]):8.2f} ms", flush=True)
        return
    dem = orc.synthetic_dem(side, side, seed=1)
    for size in (5, 7, 9, 11, 13, 15, 17, 19, 21, 33, 41, 45, 65, 81, 85):
        for route in ("direct", "mfma", "folded"):
            if route == "mfma" and size > 13:
                continue   # (more than 240 cells with taps: that form hands the call to the tap-by-tap kernel)
            if route == "direct" and size >= 45:
                route = "fft"   # (what such kernels took before the streamed form: rotated kernels of 64 cells a side and more)
            kernels = topo._valley_kernels(size, [0, 0.15, 0.3])
            taps, ksize, ang = topo._valley_ridge_tables(kernels, np.arange(0, 180, dtype=np.float32))
            os.environ["TOPO_AMD_VALLEY_MFMA_MAX_KERNEL"] = "0" if route in ("direct", "fft") else "1000"
            os.environ["TOPO_AMD_VALLEY_FOLD"] = "1" if route == "folded" else "0"
            dev = d.DeviceArray.from_host(dem)
            n, a = d.DeviceArray(side, side), d.DeviceArray(side, side)
            blk = d.Block(dev)
            ts = []
            for _ in range(4):
                d.sync()
                t0 = time.perf_counter()
                blk.valley_ridge(taps, ksize, ang, 3, 1500.0, 400.0, n, a)
                d.sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            print(f"{side}^2 size {size:2d} canvas {int(ksize.max()):2d} {route:6s}: {min(ts[1:]):8.2f} ms (route {d.valley_route()})", flush=True)
            for x in (n, a, dev):
                x.free()


if __name__ == "__main__":
    main()
