"""Time gradient / Gaussian at a few sigmas on the 32768^2 bench DEM (per-launch HIP events, median of 6)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = int(os.environ.get("N", "32768"))
sigmas = [float(a) for a in sys.argv[1:]] or [3.25, 12.0, 16.0, 22.0, 30.25]
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
o = [d.DeviceArray(n, n) for _ in range(4)]
out = {"TOPO_AMD_GAUSS_MFMA_MIN_RADIUS": os.environ.get("TOPO_AMD_GAUSS_MFMA_MIN_RADIUS"), "n": n}
for s in sigmas:
    ms = sorted(d.time_launches(lambda: blk.gradient(s, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3]), 6))
    out[f"gradient_{s}"] = round(ms[len(ms) // 2], 3)
    ms = sorted(d.time_launches(lambda: blk.gaussian(s, s, o[0]), 6))
    out[f"gaussian_{s}"] = round(ms[len(ms) // 2], 3)
print(json.dumps(out))
