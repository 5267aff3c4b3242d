"""Axis 0 of the Gaussian alone (sigma on axis 0 only), median of 7 launches (the harness of profiles/r06_gauss_axis0_s1.txt;
the lab switches it once labelled its lines with live in commit d700ebd).  usage: ax0_lab.py sigma [n]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from topo_descriptors_amd import device as d  # noqa: E402
s = float(sys.argv[1]) if len(sys.argv) > 1 else 30.25
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
o = d.DeviceArray(n, n)
ms = sorted(d.time_launches(lambda: blk.gaussian(s, 0.0, o), 7))
print(json.dumps({"sigma": s, "n": n, "axis0_ms": round(ms[3], 3), "min": round(ms[0], 3)}))
