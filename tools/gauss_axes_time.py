"""Per-axis time of the Gaussian on the 32768^2 bench DEM: sigma on axis 0 only, on axis 1 only, on both
(median of 7 launches, HIP events)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402
n = int(os.environ.get("N", "32768"))
sigmas = [float(a) for a in sys.argv[1:]] or [3.25, 30.25]
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
o = d.DeviceArray(n, n)
med = lambda f: round(sorted(d.time_launches(f, 7))[3], 3)
for s in sigmas:
    print(json.dumps({"sigma": s, "split_once": os.environ.get("TOPO_AMD_GAUSS_SPLIT_ONCE", "1"), "axis0_ms": med(lambda: blk.gaussian(s, 0.0, o)),
                      "axis1_ms": med(lambda: blk.gaussian(0.0, s, o)), "both_ms": med(lambda: blk.gaussian(s, s, o))}))
