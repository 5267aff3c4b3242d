"""Do two output planes conflict in HBM when their addresses differ by a multiple of a large power of two?  TPI + STD at
7 px (4 B read + 8 B written per pixel, HBM-like) with the STD plane placed at several byte offsets behind the TPI
plane inside one allocation; the gradient's four planes likewise.  (profiles/r03_box_spread.txt: these kernels differ
by 20 % from session to session while single-plane kernels differ by 2 %.)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d

class View:
    def __init__(self, ptr):
        self.ptr = ptr

n = 32768
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
plane = n * n * 4
extra = 64 << 20
arena = d.DeviceArray(4 * n + 4 * (extra // (n * 4)) + 64, n)   # room for 4 planes + staggers
med = lambda f: round(sorted(d.time_launches(f, 7))[3], 3)
for stagger in (0, 256, 1024, 4096, 4096 + 256, 65536, 65536 + 4096, 1 << 20, (1 << 20) + 4096 + 256, (2 << 20) + 8192, (8 << 20) + 12288):
    outs = [View(arena.ptr + k * (plane + stagger)) for k in range(4)]
    row = {"stagger_bytes": stagger}
    row["tpi_std_s7_ms"] = med(lambda: blk.tpi_std(7, tpi=outs[0], std=outs[1]))
    row["tpi_pair_7_11_ms"] = med(lambda: blk.tpi_multi([7, 11], [outs[0], outs[1]]))
    row["gradient_3.25_ms"] = med(lambda: blk.gradient(3.25, [30.0], [-30.0], dx=outs[0], dy=outs[1], slope=outs[2], aspect=outs[3]))
    print(json.dumps(row), flush=True)
