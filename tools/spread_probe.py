"""Run-to-run spread of one kernel: the axis-1 Gaussian pass at sigma 16 on the 32768^2 bench DEM reads one
plane and writes another; its time differs by up to 35 % between PROCESSES on one box.  Prints the median of 7 launches
and the device addresses of the two planes (mod 2^32) for this process; run it several times.
    python tools/spread_probe.py [sigma=16] [nx=32768]      (nx = 32832: a row pitch that is not a power of two)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
n = 32768
nx = int(sys.argv[2]) if len(sys.argv) > 2 else n
dem = d.synth_dem(n, nx, seed=0)
blk = d.Block(dem)
o = d.DeviceArray(n, nx)
med = lambda f: round(sorted(d.time_launches(f, 7))[3], 3)  # noqa: E731
print(json.dumps({"sigma": sigma, "nx": nx, "axis1_ms": med(lambda: blk.gaussian(0.0, sigma, o)), "axis0_ms": med(lambda: blk.gaussian(sigma, 0.0, o)),
                  "in_ptr_mod_4GiB": hex(dem.ptr.value & 0xffffffff) if hasattr(dem.ptr, "value") else hex(int(dem.ptr) & 0xffffffff),
                  "out_ptr_mod_4GiB": hex(o.ptr.value & 0xffffffff) if hasattr(o.ptr, "value") else hex(int(o.ptr) & 0xffffffff)}))
