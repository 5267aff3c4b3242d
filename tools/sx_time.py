"""Time topo sx on the bench DEM for several azimuths and radii (HIP events, best of 5).
usage: sx_time.py [n=32768]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
out = d.DeviceArray(n, n)
for radius in (500.0, 1000.0, 2000.0):
    for az in (0.0, 45.0, 90.0, 225.0):
        window, dj, di, dist = d.sx_offsets(az, radius, 30.0, -30.0)
        blk.sx(dj, di, dist, window, 10.0, out)
        d.sync()
        best = 1e9
        for _ in range(5):
            d.timer_start()
            blk.sx(dj, di, dist, window, 10.0, out)
            best = min(best, d.timer_stop())
        print(f"radius {radius:6.0f} m azimuth {az:5.0f}: {best:8.3f} ms  ({len(dj)} ray points)", flush=True)
