"""Error of the Gaussian against the float64 evaluation of the same filter (oracle/topo_oracle.gaussian_exact), for
the route the library takes (TOPO_AMD_GAUSS_SPLIT_ONCE=0: the tile kernels on axis 1 too).  Prints max / rms
error, where the maximum sits, and the same figures for a single axis."""
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np
from topo_descriptors_amd import topo
from oracle import topo_oracle as orc

def report(tag, dem, sigma):
    got = topo.dem(dem, sigma).astype(np.float64)
    ex = orc.gaussian_exact(dem, sigma)
    err = np.abs(got - ex)
    j, i = np.unravel_index(np.argmax(err), err.shape)
    row = {"case": tag, "sigma": sigma, "max": float(err.max()), "rms": float(np.sqrt(np.mean(err ** 2))),
           "at": [int(j), int(i)], "p99.9": float(np.quantile(err, 0.999))}
    for ax, sg in (("axis0", (sigma, 0.0)), ("axis1", (0.0, sigma))):
        e1 = np.abs(topo.dem(dem, sg).astype(np.float64) - orc.gaussian_exact(dem, sg))
        jj, ii = np.unravel_index(np.argmax(e1), e1.shape)
        row[ax] = {"max": float(e1.max()), "rms": float(np.sqrt(np.mean(e1 ** 2))), "at": [int(jj), int(ii)]}
    print(json.dumps(row))

if __name__ == "__main__":
    sigmas = [float(s) for s in sys.argv[1:]] or [2.25, 3.25, 8.0, 30.25]
    g = np.load(os.path.join("tests", "golden", "gaussian.npz"))
    for s in sigmas:
        report("golden dem_int 128x160", g["dem_int"], s)
    big = orc.synthetic_dem(1024, 1536, seed=5)
    for s in sigmas:
        report("synthetic 1024x1536", big, s)
    frac = big + np.float32(0.37) * orc.synthetic_dem(1024, 1536, seed=6) / np.float32(100.0)
    for s in sigmas:
        report("fractional 1024x1536", frac.astype(np.float32), s)
