"""Long Gaussian filters: wave-shift axis 1 against the transpose path (set
TOPO_AMD_GAUSS_WAVE_MIN_LANES=65 to force the latter).  usage: gauss_long_crossover.py [n=16384]"""
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


print("TOPO_AMD_GAUSS_WAVE_MIN_LANES =", os.environ.get("TOPO_AMD_GAUSS_WAVE_MIN_LANES", "(default)"))
for sigma in (6.1, 8.0, 12.0, 16.0, 20.0, 25.0, 30.25, 40.0, 50.0, 60.0, 75.0, 90.0, 100.0, 107.0):
    g = t(lambda: blk.gradient(sigma, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3]))
    s = t(lambda: blk.gaussian(sigma, sigma, o[0]))
    print(f"sigma {sigma:6.2f}: gradient {g:8.2f} ms, gaussian {s:8.2f} ms", flush=True)
