"""Randomised check of the multi-azimuth Sx (topo_amd_sx_multi_dev): random fans of azimuths,
radii, arcs, steps, radius_min, grid spacings, DEM shapes and row blocks; every plane must have the
bits of the single-azimuth call on the same block.   usage: fuzz_sx_multi.py [cases=300] [seed=0]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d, shard  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
planes = 0
for case in range(cases):
    ny, nx = int(rng.integers(40, 400)), int(rng.integers(40, 500))
    dem = rng.normal(1500.0, 300.0, (ny, nx)).astype(np.float32)
    if rng.random() < 0.3:
        dem = np.rint(dem)
    if rng.random() < 0.2:
        dem[rng.integers(0, ny), rng.integers(0, nx)] = np.nan
    dx = float(rng.choice([10.0, 25.0, 30.0, 50.0]))
    dy = -dx * float(rng.choice([1.0, 1.0, 0.8, 1.6]))
    radius = float(rng.choice([60.0, 150.0, 300.0, 500.0, 1000.0])) * dx / 30.0
    n_az = int(rng.integers(1, 20))
    start, step = float(rng.uniform(0, 360)), float(rng.choice([1.0, 2.5, 5.0, 10.0, 22.5, 45.0, 90.0]))
    azimuths = [(start + step * k) % 360.0 for k in range(n_az)]
    if rng.random() < 0.2:
        rng.shuffle(azimuths)
    arc = float(rng.choice([0.0, 5.0, 10.0, 30.0]))
    steps = int(rng.choice([1, 5, 15]))
    rmin = float(rng.choice([0.0, 0.0, 2.0 * dx]))
    height = float(rng.choice([0.0, 2.0, 10.0]))
    sectors = [d.sx_offsets(a, radius, dx, dy, arc, steps, rmin) for a in azimuths]
    if any(np.all(np.isnan(s[3])) for s in sectors):
        continue
    up, down = shard.sx_multi_halo(sectors)
    nblocks = int(rng.integers(1, 4))
    for row0, rows in shard.split_rows(ny, nblocks):
        lo, hi = max(0, row0 - up), min(ny, row0 + rows + down)
        dev = d.DeviceArray.from_host(dem[lo:hi])
        blk = d.Block(dev, row0=lo, gny=ny)
        outs = [d.DeviceArray(rows, nx) for _ in sectors]
        blk.sx_multi(sectors, height, outs, out_row0=row0, out_rows=rows)
        one = d.DeviceArray(rows, nx)
        for a, (window, dj, di, dist), o in zip(azimuths, sectors, outs):
            blk.sx(dj, di, dist, window, height, one, out_row0=row0, out_rows=rows)
            d.sync()
            planes += 1
            if not np.array_equal(one.to_host(), o.to_host(), equal_nan=True):
                bad += 1
                print(f"MISMATCH case {case}: {ny}x{nx} dx {dx} dy {dy} radius {radius} azimuth {a} "
                      f"arc {arc} steps {steps} rmin {rmin} rows {row0}+{rows}", flush=True)
            o.free()
        one.free()
        dev.free()
print(f"{cases} cases, {planes} planes compared, {bad} mismatches")
sys.exit(1 if bad else 0)
