"""The one failure of the round-5 fuzz run (tools/fuzz_parity.py 400 501, before the tool declared the raster class of its row
blocks): TPI alone, 19 px, a 192 x 482 window with fractional elevations, 5 row blocks.  Replays the generator to that case and
compares undeclared blocks, declared blocks and the whole raster.  usage: python tools/fuzz_repro_r05.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import topo_oracle as orc  # noqa: E402
from topo_descriptors_amd import _lib, device as d, shard, topo  # noqa: E402

ny, nx, size, nb = 192, 482, 19, 5
for seed in range(40):
    rng = np.random.default_rng(seed)
    dem = (orc.synthetic_dem(ny, nx, seed=seed, row0=int(rng.integers(0, 5000)), col0=int(rng.integers(0, 5000))) +
           rng.random((ny, nx))).astype(np.float32)
    whole = topo.tpi(dem, size)
    up, down = shard.halo_rows(_lib.DESC_TPI, size)
    out = {}
    for declared in (False, True):
        scan = d.RasterScan()
        if declared:
            for row0, rows in shard.split_rows(ny, nb):
                dev = d.DeviceArray.from_host(dem[row0:row0 + rows])
                scan.add(d.Block(dev, row0=row0, gny=ny))
                dev.free()
        parts = []
        for row0, rows in shard.split_rows(ny, nb):
            lo, hi = max(0, row0 - up), min(ny, row0 + rows + down)
            dev = d.DeviceArray.from_host(dem[lo:hi])
            blk = d.Block(dev, row0=lo, gny=ny)
            if declared:
                scan.declare(blk)
            c = d.DeviceArray(rows, nx)
            blk.tpi_std(size, tpi=c, out_row0=row0, out_rows=rows)
            d.sync()
            parts.append(c.to_host())
            c.free(), dev.free()
        out[declared] = int((np.concatenate(parts) != whole).sum())
    print(f"seed {seed}: range {float(dem.max() - dem.min()):7.1f} m   pixels differing from the whole raster: undeclared blocks {out[False]}, declared blocks {out[True]}")
