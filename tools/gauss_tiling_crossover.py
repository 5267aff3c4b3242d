"""Gaussian / gradient around the switch from the narrow (8 x 8) to the wide (16 x 16) register
blocking and, for the gradient, from the LDS-tiled fused axis 1 to the wave-shift one
(TOPO_AMD_GAUSS_WIDE_MIN_RADIUS, radius = int(4 sigma + 0.5)).  usage: gauss_tiling_crossover.py [n=16384]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dem = d.synth_dem(n, n, seed=0)
o = [d.DeviceArray(n, n) for _ in range(4)]
blk = d.Block(dem)


def t(fn):
    fn()
    d.sync()
    d.timer_start()
    fn()
    fn()
    return d.timer_stop() / 2


print("TOPO_AMD_GAUSS_WIDE_MIN_RADIUS =", os.environ.get("TOPO_AMD_GAUSS_WIDE_MIN_RADIUS", "(default)"))
for sigma in (4.0, 5.75, 6.0, 7.0, 8.0, 10.0, 12.0, 14.0, 16.0, 20.0, 23.0):
    g = t(lambda: blk.gradient(sigma, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3]))
    s = t(lambda: blk.gaussian(sigma, sigma, o[0]))
    print(f"sigma {sigma:5.2f} radius {int(4 * sigma + 0.5):3d}: gradient {g:7.2f} ms, gaussian {s:7.2f} ms", flush=True)
