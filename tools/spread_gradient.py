"""Run-to-run spread of the gradient: median of 7 launches of the four-output gradient at sigma 3.25
and 30.25 on the 32768^2 bench DEM, one line per process; run it several times."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = 32768
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
o = [d.DeviceArray(n, n) for _ in range(4)]
med = lambda f: round(sorted(d.time_launches(f, 7))[3], 3)  # noqa: E731
print(json.dumps({s: med(lambda: blk.gradient(s, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3])) for s in (3.25, 30.25)}))
