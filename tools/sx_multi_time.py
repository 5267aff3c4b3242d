"""Sx for a fan of azimuths: one multi-sector call against a loop of single calls (HIP events,
best of 3).  usage: sx_multi_time.py [n=32768]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
outs = [d.DeviceArray(n, n) for _ in range(8)]
for radius, step in ((500.0, 5.0), (500.0, 10.0), (500.0, 45.0), (1000.0, 5.0), (2000.0, 5.0)):
    azimuths = [step * k for k in range(8)]
    sectors = [d.sx_offsets(a, radius, 30.0, -30.0) for a in azimuths]

    def loop():
        for (window, dj, di, dist), out in zip(sectors, outs):
            blk.sx(dj, di, dist, window, 10.0, out)

    def multi():
        blk.sx_multi(sectors, 10.0, outs)

    times = {}
    for name, fn in (("loop", loop), ("multi", multi)):
        fn()
        d.sync()
        best = 1e9
        for _ in range(3):
            d.timer_start()
            fn()
            best = min(best, d.timer_stop())
        times[name] = best
    print(f"radius {radius:6.0f} m, 8 azimuths every {step:4.1f} deg: loop {times['loop']:8.3f} ms, "
          f"one call {times['multi']:8.3f} ms ({times['loop'] / times['multi']:.2f}x)", flush=True)
