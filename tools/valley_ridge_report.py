"""Parity figures and timings of the valley / ridge index (test-side tool: imports the oracle).
Run on the GPU box: python tools/valley_ridge_report.py"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import topo_oracle as orc  # noqa: E402
from topo_descriptors_amd import device as d, topo  # noqa: E402

g = np.load(os.path.join(REPO, "tests", "golden", "valley_ridge.npz"))
print("case                      max|ref|  gpu-ref  (rel)     gpu-exact (rel)    ref-exact  dir==ref")
for tag in ["int_valley_s7", "int_ridge_s7", "int_valley_s5", "int_valley_s17", "int_valley_s9_flat0",
            "int_ridge_s9_flat2", "frac_valley_s7", "frac_valley_s9_sig"]:
    p = g[f"{tag}_params"]
    size, mode, sigma, flats = int(p[0]), ("valley", "ridge")[int(p[1])], (None if p[2] < 0 else float(p[2])), list(p[3:])
    dem = g["dem_int"] if tag.startswith("int") else g["dem_frac"]
    norm, direction = topo.valley_ridge(dem, size, mode, flats, sigma)
    ex = orc.valley_ridge_exact(dem, size, mode, flats, sigma)[0]
    ref, dref = g[f"{tag}_norm"], g[f"{tag}_dir"]
    s = float(np.max(np.abs(ref)))
    a, b, c = float(np.max(np.abs(norm - ref))), float(np.max(np.abs(norm - ex))), float(np.max(np.abs(ref - ex)))
    print(f"{tag:24s} {s:9.3f} {a:9.2e} ({a / s:8.1e}) {b:9.2e} ({b / s:8.1e}) {c:9.2e}  {np.mean(direction == dref):.4f}")

# timings, device-resident, HIP events
out = {}
for n, sizes in ((8192, (7, 17)), (32768, (7,))):
    dem = d.synth_dem(n, n, seed=0)
    mean, stdev = d.mean_std(dem)
    blk = d.Block(dem)
    o1, o2 = d.DeviceArray(n, n), d.DeviceArray(n, n)
    for size in sizes:
        kernels = topo._valley_kernels(size, [0, 0.15, 0.3])
        t0 = time.perf_counter()
        taps, ksize, angles = topo._valley_ridge_tables(kernels, np.arange(0, 180, dtype=np.float32))
        host_s = time.perf_counter() - t0
        blk.valley_ridge(taps, ksize, angles, 3, mean, stdev, o1, o2)
        d.sync()
        d.timer_start()
        blk.valley_ridge(taps, ksize, angles, 3, mean, stdev, o1, o2)
        ms = d.timer_stop()
        ntaps = int((ksize.astype(np.int64) ** 2).sum())
        nonzero = int(np.any(taps.reshape(-1, 4)[:, :3] != 0, axis=1).sum())  # what the kernel evaluates
        out[f"{n}x{n}_s{size}"] = {"ms": round(ms, 2), "Mpixels_per_s": round(n * n / ms / 1e3, 1),
                                   "taps_over_180_angles": ntaps, "nonzero_taps": nonzero, "planes": 3,
                                   "GFMA_per_s_executed": round(n * n * nonzero * 3 / ms / 1e6, 0),
                                   "host_kernel_tables_s": round(host_s, 3)}
    for a in (o1, o2, dem):
        a.free()
# the reference's path on one host core, bounded sample
sample = orc.synthetic_dem(1024, 1024, seed=0)
t0 = time.perf_counter()
orc.valley_ridge_scipy(sample, 7, "valley")
dt = time.perf_counter() - t0
out["cpu_reference_path_1_core"] = {"sample": "1024x1024, size 7, oracle.valley_ridge_scipy (the reference's FFT calls)",
                                    "s": round(dt, 2), "Mpixels_per_s": round(1024 * 1024 / dt / 1e6, 4)}
print(json.dumps(out, indent=1))
