"""Gaussian / gradient on a DEM whose width is not a multiple of 4 (32768 rows x 32765 columns): the matrix-core routes
take any width since round 3 (the switch back to the vector-ALU kernels, TOPO_AMD_GAUSS_MFMA_ANY_WIDTH, is gone)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d
ny, nx = 32768, 32765
dem = d.synth_dem(ny, nx, seed=0)
blk = d.Block(dem)
o = [d.DeviceArray(ny, nx) for _ in range(4)]
med = lambda f: round(sorted(d.time_launches(f, 5))[2], 3)
row = {"nx": nx}
for s in (3.25, 30.25):
    row[f"gaussian_{s}"] = med(lambda: blk.gaussian(s, s, o[0]))
    row[f"gradient_{s}"] = med(lambda: blk.gradient(s, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3]))
print(json.dumps(row))
