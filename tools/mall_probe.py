"""Does the 256 MB Infinity Cache serve a write -> read hand-over between kernels?  Gaussian (two passes) and
gradient on DEMs of 32768 columns and fewer and fewer rows: ns per pixel.  Below ~60 MB per plane everything a
launch touches fits the cache."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402
nx = 32768
for ny in (32768, 8192, 2048, 1024, 512, 256):
    dem = d.synth_dem(ny, nx, seed=0)
    blk = d.Block(dem)
    o = [d.DeviceArray(ny, nx) for _ in range(4)]
    med = lambda f: sorted(d.time_launches(f, 9))[4]
    row = {"rows": ny, "MB_per_plane": round(ny * nx * 4 / 1e6, 1)}
    for s in (3.25,):
        g = med(lambda: blk.gaussian(s, s, o[0]))
        gr = med(lambda: blk.gradient(s, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3]))
        row[f"gaussian_{s}_ns_per_px"] = round(g * 1e6 / (ny * nx), 4)
        row[f"gradient_{s}_ns_per_px"] = round(gr * 1e6 / (ny * nx), 4)
        row[f"gaussian_{s}_ms"] = round(g, 3)
        row[f"gradient_{s}_ms"] = round(gr, 3)
    print(json.dumps(row))
    for a in o:
        a.free()
    blk.free() if hasattr(blk, "free") else None
    dem.free()
