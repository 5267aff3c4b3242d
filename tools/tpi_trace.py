"""Launch topo tpi at 67 px on the bench DEM a few times: the workload for `rocprofv3 --kernel-trace` /
tools/pmc_passes.sh when looking at the TPI kernels.  python tools/tpi_trace.py [n=32768] [size=67] [int|frac|both=int]
(int: whole metres - what the traffic and vector-ALU-bound files of the headline are taken on; frac: fractional
elevations - the first call starts with the whole-metre kernel, the later ones with the scaled take-all kernel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
size = int(sys.argv[2]) if len(sys.argv) > 2 else 67
t = d.DeviceArray(n, n)
which = sys.argv[3] if len(sys.argv) > 3 else "int"
for integer in {"int": (True,), "frac": (False,), "both": (True, False)}[which]:
    dem = d.synth_dem(n, n, seed=0, integer=integer)
    blk = d.Block(dem)
    for _ in range(6):
        blk.tpi_std(size, tpi=t)
    d.sync()
    dem.free()
