"""Tiny and odd shapes through the Gaussian / gradient routes against scipy: every (rows, columns) in a grid with
rows 1 ... 70 and columns 4 ... 132 (multiples of 4: the matrix-core routes; others: the vector-ALU kernels),
sigmas that pick the fused kernel (both raw-block widths), the two-pass f16 kernels and the widest filter, plus
huge, infinite and NaN samples.  Prints the worst error per sigma and any mismatch of the non-finite masks."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import ndimage
from topo_descriptors_amd import topo
rng = np.random.default_rng(3)
rows_list = [1, 2, 3, 5, 16, 31, 32, 33, 47, 64, 65, 70]
cols_list = [4, 8, 12, 28, 32, 36, 60, 64, 68, 96, 128, 132, 30, 67]
bad = 0
for sigma in (1.0, 3.25, 6.0, 10.0, 13.0, 20.0, 30.25):
    worst = 0.0
    for ny, nx in itertools.product(rows_list, cols_list):
        dem = (1500 + 300 * rng.standard_normal((ny, nx))).astype(np.float32)
        got = topo.dem(dem, sigma)
        ref = ndimage.gaussian_filter(dem, sigma)
        err = float(np.max(np.abs(got - ref)))
        worst = max(worst, err)
        if err > 2e-3:
            bad += 1
            print("VALUE", sigma, ny, nx, err)
        if ny >= 5 and nx >= 8:
            d2 = dem.copy()
            d2[ny // 2, nx // 3] = np.nan
            d2[ny - 1, nx - 1] = np.inf
            d2[0, 0] = 3.0e9
            g2 = topo.dem(d2, sigma)
            r2 = ndimage.gaussian_filter(d2, sigma)
            if not np.array_equal(np.isfinite(g2), np.isfinite(r2)):
                if nx % 4 == 0:  # the matrix-core routes: exactly scipy's footprint
                    bad += 1
                    print("MASK", sigma, ny, nx, int(np.sum(np.isfinite(g2) != np.isfinite(r2))))
                elif np.any(np.isfinite(g2) & ~np.isfinite(r2)):  # vector-ALU kernels: a superset (padded tap chunks)
                    bad += 1
                    print("MASK-subset", sigma, ny, nx)
            else:
                fin = np.isfinite(r2)
                scale = max(1.0, float(np.max(np.abs(r2[fin])))) if fin.any() else 1.0
                if fin.any() and float(np.max(np.abs(g2[fin] - r2[fin]))) > 2e-6 * scale + 2e-3:
                    bad += 1
                    print("VALUE-with-wild", sigma, ny, nx, float(np.max(np.abs(g2[fin] - r2[fin]))), scale)
    print(f"sigma {sigma}: worst |gpu - scipy| {worst:.2e} over {len(rows_list) * len(cols_list)} shapes", flush=True)
print("failures:", bad)
