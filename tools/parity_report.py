#!/usr/bin/env python3
"""Print the parity table: GPU vs reference golden, GPU vs exact float64, reference floor.
Test-side tool (imports the oracle); run on the GPU box: python tools/parity_report.py"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import topo_oracle as orc  # noqa: E402
from topo_descriptors_amd import topo  # noqa: E402

G = os.path.join(REPO, "tests", "golden")


def load(name):
    with np.load(os.path.join(G, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def row(name, got, ref, exact, aspect=False, mask=None):
    got = np.asarray(got, np.float64)
    if aspect:
        d_ref = orc.wrapped_angle_diff(got, ref)
        d_ex = orc.wrapped_angle_diff(got, exact)
        d_fl = orc.wrapped_angle_diff(ref, exact)
        scale = 360.0
    else:
        d_ref, d_ex, d_fl = np.abs(got - ref), np.abs(got - exact), np.abs(ref - exact)
        scale = np.nanmax(np.abs(ref))
    if mask is not None:
        d_ref, d_ex, d_fl = d_ref[mask], d_ex[mask], d_fl[mask]
    print(f"{name:28s} max|ref|={scale:10.4g}  gpu-ref={np.nanmax(d_ref):9.3g} ({np.nanmax(d_ref)/scale:8.2e})"
          f"  gpu-exact={np.nanmax(d_ex):9.3g} ({np.nanmax(d_ex)/scale:8.2e})  ref-exact={np.nanmax(d_fl):9.3g}")


def main():
    g = load("tpi_std")
    for tag in ("int", "frac"):
        dem = g["dem_" + tag]
        for size in (3, 5, 6, 7, 17, 65):
            row(f"tpi_{tag}_s{size}", topo.tpi(dem, size), g[f"tpi_{tag}_s{size}"], orc.tpi_exact(dem, size))
        for size in (3, 5, 6, 7, 17, 65):
            row(f"std_{tag}_s{size}", topo.std(dem, size), g[f"std_{tag}_s{size}"], orc.std_exact(dem, size))
        row(f"tpi_{tag}_s7_sig", topo.tpi(dem, 7, sigma=1.75), g[f"tpi_{tag}_s7_sig1p75"], orc.tpi_exact(dem, 7, sigma=1.75))
        row(f"std_{tag}_s17_sig", topo.std(dem, 17, sigma=2.125), g[f"std_{tag}_s17_sig2p125"], orc.std_exact(dem, 17, sigma=2.125))
    g = load("gaussian")
    for key, src in (("gauss_int_0.75", "dem_int"), ("gauss_int_2.25", "dem_int"), ("gauss_int_3.25", "dem_int"),
                     ("gauss_big_30.25", "dem_big"), ("gauss_small_8.0", "dem_small")):
        sigma = float(key.rsplit("_", 1)[1])
        row(key, topo.dem(g[src], sigma), g[key], orc.gaussian_exact(g[src], sigma))
    g = load("gradient")
    cases = [("sob_n", 0.75, "n", 1, "dem_int"), ("g3_n", 3.25, "n", 1, "dem_int"), ("g3_s", 3.25, "s", 1, "dem_int"),
             ("g3_2d", 3.25, "2d", 1, "dem_int"), ("g3_r2_n", 3.25, "n", 2, "dem_int"),
             ("g2_r05_n", 2.25, "n", 0.5, "dem_int"), ("g30_big", 30.25, "b", 1, "dem_big")]
    for tag, sigma, rt, ratio, src in cases:
        res = {"x": g[f"res_{rt}_x"], "y": g[f"res_{rt}_y"]}
        got = topo.gradient(g[src], sigma, res, sig_ratio=ratio)
        ex = orc.gradient_exact(g[src], sigma, res, sig_ratio=ratio)
        steep = g[f"{tag}_slope"] > 0.1
        for k, nm in enumerate(("dx", "dy", "slope", "aspect")):
            row(f"{tag}_{nm}", got[k], g[f"{tag}_{nm}"], ex[k], aspect=(nm == "aspect"),
                mask=steep if nm == "aspect" else None)


if __name__ == "__main__":
    main()
