"""The fused matrix-core axis-1 + epilogue kernel (gauss_axis1_mfma_grad_kernel) against the unfused route
(TOPO_AMD_GRAD_FUSED_MFMA=0), bit for bit, over shapes, sigmas, resolutions, output subsets and row blocks; two child
processes (the switch is read once per process).  Also ms per call at 32768^2."""
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(420, 320, 3.25), (300, 512, 2.25), (333, 516, 3.9), (1000, 2048, 3.25), (5000, 1024, 2.0), (64, 64, 3.25),
         (31, 128, 3.25), (9000, 256, 3.25), (2100, 4096, 4.0)]

if len(sys.argv) > 1:
    os.environ["TOPO_AMD_GRAD_FUSED_MFMA"] = sys.argv[1]
    sys.path.insert(0, REPO)
    from oracle import topo_oracle as orc
    from topo_descriptors_amd import device as d
    out = {}
    for n, (gny, nx, sigma) in enumerate(CASES):
        dem = orc.synthetic_dem(gny, nx, seed=n)
        if n == 1:
            dem[100, 200] = np.nan
        dev = d.DeviceArray.from_host(dem)
        blk = d.Block(dev)
        x = 2600000.0 + 30.0 * np.arange(nx)
        y = 1200000.0 - 30.0 * np.arange(gny)
        res = orc.grid_resolution(x, y)
        o = [d.DeviceArray(gny, nx) for _ in range(4)]
        blk.gradient(sigma, res["x"], res["y"], dx=o[0], dy=o[1], slope=o[2], aspect=o[3])
        d.sync()
        out[f"c{n}_full"] = np.stack([a.to_host() for a in o])
        blk.gradient(sigma, [30.0], [-30.0], slope=o[2], aspect=o[3])  # scalar resolution, two planes only
        d.sync()
        out[f"c{n}_sa"] = np.stack([o[2].to_host(), o[3].to_host()])
        if gny >= 300:  # a row block in the middle, with the ghost rows the library asks for
            r0, rows = gny // 3 + 7, gny // 3
            from topo_descriptors_amd import _lib, shard
            up, down = shard.halo_rows(_lib.DESC_GRADIENT, sigma, 1.0)
            lo, hi = max(0, r0 - up), min(gny, r0 + rows + down)
            part = d.DeviceArray.from_host(dem[lo:hi])
            po = [d.DeviceArray(rows, nx) for _ in range(4)]
            d.Block(part, row0=lo, gny=gny).gradient(sigma, res["x"], res["y"], dx=po[0], dy=po[1], slope=po[2], aspect=po[3],
                                                     out_row0=r0, out_rows=rows)
            d.sync()
            blockp = np.stack([a.to_host() for a in po])
            assert np.array_equal(blockp, out[f"c{n}_full"][:, r0:r0 + rows], equal_nan=True), ("row block", n)
            for a in po + [part]:
                a.free()
        if sys.argv[1] == "1" and gny <= 1000:
            exact = orc.gradient_exact(dem, sigma, res)
            for k in range(3):
                err = np.nanmax(np.abs(out[f"c{n}_full"][k] - exact[k])) / np.nanmax(np.abs(exact[k]))
                assert err <= 1e-4 or n == 1, (n, k, err)
        for a in o + [dev]:
            a.free()
    np.savez(sys.argv[2], **out)
    if len(sys.argv) > 3:
        n = 32768
        dem = d.synth_dem(n, n, seed=0)
        blk = d.Block(dem)
        o = [d.DeviceArray(n, n) for _ in range(4)]
        for sigma in (2.25, 3.25):
            ms = sorted(d.time_launches(lambda: blk.gradient(sigma, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3]), 6))
            ms2 = sorted(d.time_launches(lambda: blk.gradient(sigma, [30.0], [-30.0], slope=o[2], aspect=o[3]), 6))
            print(f"TOPO_AMD_GRAD_FUSED_MFMA={sys.argv[1]} sigma {sigma}: four outputs {ms[3]:.3f} ms, slope+aspect {ms2[3]:.3f} ms")
else:
    for m, f in (("1", "/tmp/g_fused.npz"), ("0", "/tmp/g_unfused.npz")):
        subprocess.check_call([sys.executable, __file__, m, f, "time"])
    a, b = np.load("/tmp/g_fused.npz"), np.load("/tmp/g_unfused.npz")
    bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
    print("fused == unfused on", len(a.files), "planes sets:", not bad, bad)
    for k in bad:
        diff = a[k] != b[k]
        diff &= ~(np.isnan(a[k]) & np.isnan(b[k]))
        idx = np.argwhere(diff)
        print(k, "differing:", int(diff.sum()), "first", idx[:5].tolist(), "planes", sorted(set(idx[:, 0].tolist())),
              "rows", idx[:, 1].min(), idx[:, 1].max(), "cols", idx[:, 2].min(), idx[:, 2].max())
