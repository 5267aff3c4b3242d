"""The calls of the reference's example script (scripts/compute_topo_descriptors.py) with its own 12
scales (100 m ... 100 km) on a synthetic DEM at 100 m spacing, timed call by call (wall clock,
results returned as host arrays, nothing written to disk).  usage: example_script_timing.py [n=8192]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import batch, device as d, helpers as hlp  # noqa: E402


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


n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = d.synth_dem(n, n, seed=0)
dem = dev.to_host()
dev.free()
ds = Dataset(dem, 2600000.0 + 100.0 * np.arange(n), 1200000.0 - 100.0 * np.arange(n))
scales = [100, 300, 500, 1000, 2000, 4000, 6000, 10000, 20000, 30000, 60000, 100000]
print(f"{n}x{n} at 100 m; disc / kernel sizes in pixels: {list(hlp.scale_to_pixel(scales, ds)[0])}", flush=True)
ind_nans = (np.array([5]), np.array([7]))
total = 0.0


def run(label, fn):
    global total
    t0 = time.perf_counter()
    out = fn()
    dt = time.perf_counter() - t0
    total += dt
    print(f"{label:58s} {dt:8.2f} s  ({len(out)} arrays)", flush=True)


run("compute_dem, 12 scales", lambda: batch.compute_dem(ds, scales, ind_nans=ind_nans, outdir=None))
run("compute_tpi, 12 scales", lambda: batch.compute_tpi(ds, scales, smth_factors=None, ind_nans=ind_nans, outdir=None))
run("compute_tpi with prior smoothing, 12 scales", lambda: batch.compute_tpi(ds, scales, smth_factors=1, ind_nans=ind_nans, outdir=None))
run("compute_gradient, 12 scales", lambda: batch.compute_gradient(ds, scales, sig_ratios=1, ind_nans=ind_nans, outdir=None))
run("compute_std, 12 scales", lambda: batch.compute_std(ds, scales, ind_nans=ind_nans, outdir=None))
run("compute_valley_ridge valley, 9 scales (1 km ... 100 km)",
    lambda: batch.compute_valley_ridge(ds, scales[3:], mode="valley", flat_list=[0, 0.2, 0.4], smth_factors=0.5,
                                       ind_nans=ind_nans, outdir=None))
run("compute_valley_ridge ridge, 9 scales",
    lambda: batch.compute_valley_ridge(ds, scales[3:], mode="ridge", flat_list=[0, 0.15, 0.3], smth_factors=0.5,
                                       ind_nans=ind_nans, outdir=None))
run("compute_sx azimuth 0, radius 1000 m", lambda: batch.compute_sx(ds, 0, 1000, outdir=None))
print(f"{'total':58s} {total:8.2f} s")
