"""Row-block form of the C ABI on one GPU: a DEM cut into row blocks with ghost rows must give
results bit-identical to the single-block run (last row), and the
full-size configurations of BASELINE.json must agree with the oracle on sampled windows."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import topo_oracle as orc

pytestmark = pytest.mark.gpu

from topo_descriptors_amd import _lib, device as d, shard  # noqa: E402


def run_blocks(dem, nblocks, above, below, call):
    """Evaluate `call(block, out_row0, out_rows, outs)` on each of `nblocks` row blocks (each
    uploaded separately with exactly its ghost rows) and stitch the outputs."""
    gny, nx = dem.shape
    pieces = None
    for row0, rows in shard.split_rows(gny, nblocks):
        lo = max(0, row0 - above)
        hi = min(gny, row0 + rows + below)
        dev = d.DeviceArray.from_host(dem[lo:hi])
        blk = d.Block(dev, row0=lo, gny=gny)
        outs = call(blk, row0, rows)
        d.sync()
        host = [o.to_host() for o in outs]
        pieces = [[h] for h in host] if pieces is None else [p + [h] for p, h in zip(pieces, host)]
        for o in outs:
            o.free()
        dev.free()
    return [np.concatenate(p, axis=0) for p in pieces]


def halo(desc, p0, p1=0.0):
    return shard.halo_rows(desc, p0, p1)


def tpi_alone_bound(size, integer=False):
    """How far topo.tpi may be from the exact TPI where a disc holds fractional elevations: from 19 px on (the marching
    kernels) such tiles sum x in units of 2^-8 m (csrc/disc_wave_impl.hpp, tpi_scaled_march_kernel): at most 2^-9 m per
    sample, hence on the mean and on TPI.  The fused TPI + STD route and the smaller discs stay exact (2^-16 m)."""
    scaled = (not integer) and size % 2 == 1 and 19 <= size <= 101 and os.environ.get("TOPO_AMD_TPI_FRACTION_EXACT", "0") == "0"
    return 2.0 ** -9 if scaled else 0.0


@pytest.mark.parametrize("size", [6, 7, 17, 67])
@pytest.mark.parametrize("nx", [256, 250])   # 250: nx % 4 != 0 -> generic kernel
@pytest.mark.parametrize("integer", [True, False])
def test_tpi_std_blocks_bit_identical(size, nx, integer):
    dem = orc.synthetic_dem(400, nx, seed=size, integer=integer)
    up, down = halo(_lib.DESC_TPI, size)

    def call(blk, row0, rows):
        t, s = d.DeviceArray(rows, nx), d.DeviceArray(rows, nx)
        blk.tpi_std(size, tpi=t, std=s, out_row0=row0, out_rows=rows)
        return [t, s]

    def call_tpi_only(blk, row0, rows):
        t = d.DeviceArray(rows, nx)
        blk.tpi_std(size, tpi=t, out_row0=row0, out_rows=rows)
        return [t]

    whole = run_blocks(dem, 1, up, down, call)
    alone = run_blocks(dem, 1, up, down, call_tpi_only)[0]
    bound = tpi_alone_bound(size, integer)
    if bound == 0.0:
        assert np.array_equal(alone, whole[0]), size  # one exact pipeline behind both entry points
    else:
        assert np.max(np.abs(alone - whole[0])) <= bound, size
    for nb in (2, 3):
        parts = run_blocks(dem, nb, up, down, call)
        assert np.array_equal(parts[0], whole[0]), (size, nb, "tpi")
        assert np.array_equal(parts[1], whole[1]), (size, nb, "std")
        assert np.array_equal(run_blocks(dem, nb, up, down, call_tpi_only)[0], alone), (size, nb)
    assert np.max(np.abs(whole[0] - orc.tpi_exact(dem, size))) <= 2.5e-4
    e = orc.std_exact(dem, size)
    assert np.max(np.abs(whole[1] - e)) <= 1e-4 * np.max(e)


# radius int(4 sigma + 0.5): 13, 28 and 48 (fused LDS-tiled axis 1, 3 / 3 / 4 samples per lane held for
# the next tile), 64 and 88 (the same kernel with 16-wide tap chunks, 4 / 5 samples), 104 (wave-shift axis 1)
@pytest.mark.parametrize("sigma", [0.75, 3.25, 7.0, 10.0, 12.0, 16.0, 22.0, 26.0])
def test_gradient_blocks_bit_identical(sigma):
    gny, nx = 420, 320
    dem = orc.synthetic_dem(gny, nx, seed=9)
    x = 2600000.0 + 30.0 * np.arange(nx)
    y = 1200000.0 - 30.0 * np.arange(gny)
    res = orc.grid_resolution(x, y)
    up, down = halo(_lib.DESC_GRADIENT, sigma)

    def call(blk, row0, rows):
        outs = [d.DeviceArray(rows, nx) for _ in range(4)]
        blk.gradient(sigma, res["x"], res["y"], dx=outs[0], dy=outs[1], slope=outs[2], aspect=outs[3],
                     out_row0=row0, out_rows=rows)
        return outs

    whole = run_blocks(dem, 1, up, down, call)
    for nb in (2, 3):
        parts = run_blocks(dem, nb, up, down, call)
        for k in range(4):
            assert np.array_equal(parts[k], whole[k]), (sigma, nb, k)
    exact = orc.gradient_exact(dem, sigma, res)
    for k in range(3):
        assert np.max(np.abs(whole[k] - exact[k])) <= 1e-4 * np.max(np.abs(exact[k]))


@pytest.mark.parametrize("sigma", [3.25, 6.0, 10.0, 16.0])
def test_gradient_blocks_with_non_finite_samples(sigma):
    """The smooth of the gradient at radius 4 ... 47 is ONE kernel that raises a flag when it meets a sample that is
    not a plain finite one; the two-pass kernels queued behind it then redo the block.  A block without such a sample
    keeps the fused kernel's result: the two routes must give the same bits, and the non-finite footprint must not
    depend on the cut either (sigma 16: the two-pass kernels alone)."""
    gny, nx = 420, 320
    dem = orc.synthetic_dem(gny, nx, seed=19)
    dem[150, 100] = np.nan
    dem[333, 31:34] = np.inf
    up, down = halo(_lib.DESC_GRADIENT, sigma)

    def call(blk, row0, rows):
        outs = [d.DeviceArray(rows, nx) for _ in range(4)]
        blk.gradient(sigma, [30.0], [-30.0], dx=outs[0], dy=outs[1], slope=outs[2], aspect=outs[3], out_row0=row0, out_rows=rows)
        return outs

    whole = run_blocks(dem, 1, up, down, call)
    assert np.isnan(whole[0][150, 100]) and np.isfinite(whole[0][10, 10])
    for nb in (2, 3, 5):
        parts = run_blocks(dem, nb, up, down, call)
        for k in range(4):
            assert np.array_equal(parts[k], whole[k], equal_nan=True), (sigma, nb, k)


@pytest.mark.parametrize("azimuth", [0.0, 135.0, 260.0])
def test_sx_blocks_bit_identical(azimuth):
    gny, nx = 300, 260
    dem = orc.synthetic_dem(gny, nx, seed=13)
    window, dj, di, dist = d.sx_offsets(azimuth, 500.0, 30.0, -30.0)
    up, down = halo(_lib.DESC_SX, max(0, -dj.min()), max(0, dj.max()))

    def call(blk, row0, rows):
        out = d.DeviceArray(rows, nx)
        blk.sx(dj, di, dist, window, 10.0, out, out_row0=row0, out_rows=rows)
        return [out]

    whole = run_blocks(dem, 1, up, down, call)[0]
    for nb in (2, 3):
        assert np.array_equal(run_blocks(dem, nb, up, down, call)[0], whole), (azimuth, nb)
    x = 2600000.0 + 30.0 * np.arange(nx)
    y = 1200000.0 - 30.0 * np.arange(gny)
    want = orc.sx(dem, x, y, azimuth, 500.0)
    assert np.max(np.abs(whole - want)) <= 1e-4 * np.max(np.abs(want))


def test_single_rank_shard_entry_points():
    """topo_amd_shard_* with one rank: the exchange is a no-op, interior/seam split still runs."""
    gny, nx, size = 300, 256, 17
    dem = orc.synthetic_dem(gny, nx, seed=5)
    up, down = halo(_lib.DESC_TPI, size)
    plan = shard.RowShardPlan(gny, nx, 1, 0, up, down)
    sd = shard.ShardedDEM(plan, dem)
    t, s = d.DeviceArray(gny, nx), d.DeviceArray(gny, nx)
    sd.tpi_std(size, tpi=t, std=s)
    d.sync()
    assert np.max(np.abs(t.to_host() - orc.tpi_exact(dem, size))) <= 2.5e-4
    e = orc.std_exact(dem, size)
    assert np.max(np.abs(s.to_host() - e)) <= 1e-4 * np.max(e)
    window, dj, di, dist = d.sx_offsets(0.0, 300.0, 30.0, -30.0)
    plan = shard.RowShardPlan(gny, nx, 1, 0, max(0, -dj.min()), max(0, dj.max()))
    sd2 = shard.ShardedDEM(plan, dem)
    o = d.DeviceArray(gny, nx)
    sd2.sx(dj, di, dist, window, 10.0, o)
    d.sync()
    x = 2600000.0 + 30.0 * np.arange(nx)
    y = 1200000.0 - 30.0 * np.arange(gny)
    want = orc.sx(dem, x, y, 0.0, 300.0)
    assert np.max(np.abs(o.to_host() - want)) <= 1e-4 * np.max(np.abs(want))


# ---- BASELINE.json configurations at full size, checked on sampled windows ---------------------
def check_aspect(got, exact, dxy_tol, tag):
    """Wrapped aspect against the float64 evaluation where slope > 0.1 deg (tolerance contract).  Aspect is
    atan2(dx, dy): an error e in (dx, dy) moves it by up to atan(e / |grad|), which at slope 0.1 deg (|grad| 1.7e-3) is
    far more than 1e-4 x 360 deg, so the bound is the contract's 0.036 deg plus what the dx / dy tolerance allows at
    each pixel's own gradient length."""
    dx, dy, slope, aspect = exact
    steep = slope > 0.1
    assert np.all((got >= 0) & (got < 360)), tag
    if not np.any(steep):
        return
    g = np.sqrt(dx.astype(np.float64) ** 2 + dy.astype(np.float64) ** 2)
    allowed = 0.036 + np.degrees(np.arctan2(np.sqrt(2.0) * dxy_tol, g))
    diff = orc.wrapped_angle_diff(got, aspect)
    assert np.all(diff[steep] <= allowed[steep]), (tag, float(np.max((diff - allowed)[steep])))


def windows(gny, nx, side, n, seed, margin):
    rng = np.random.default_rng(seed)
    out = [(0, 0), (gny - side, nx - side), (0, nx - side)]  # corners see the boundary rules
    for _ in range(n):
        out.append((int(rng.integers(margin, gny - side - margin)),
                    int(rng.integers(margin, nx - side - margin))))
    return out


def test_config2_tpi_std_8192():
    """configs[1]: 8192 x 8192, TPI + STD at 7 and 65 px."""
    n = 8192
    dev = d.synth_dem(n, n, seed=1)
    dem = dev.to_host()
    blk = d.Block(dev)
    t, s = d.DeviceArray(n, n), d.DeviceArray(n, n)
    for size in (7, 65):
        blk.tpi_std(size, tpi=t, std=s)
        d.sync()
        th, sh = t.to_host(), s.to_host()
        r = size
        for (j, i) in windows(n, n, 256, 4, size, r):
            j0, j1, i0, i1 = max(0, j - r), min(n, j + 256 + r), max(0, i - r), min(n, i + 256 + r)
            sub = dem[j0:j1, i0:i1]
            # windows cut inside the DEM: compare away from the cut edges only
            a, b = j - j0, i - i0
            want_t = orc.tpi_exact(sub, size)[a:a + 256, b:b + 256]
            want_s = orc.std_exact(sub, size)[a:a + 256, b:b + 256]
            got_t = th[j:j + 256, i:i + 256]
            got_s = sh[j:j + 256, i:i + 256]
            # (every window is cut r = size >= size // 2 pixels outside its 256 x 256 core or at a true DEM edge, so
            # no disc of the core crosses a cut)
            assert np.max(np.abs(got_t - want_t)) <= 2.5e-4, (size, j, i)
            assert np.max(np.abs(got_s - want_s)) <= 1e-4 * max(np.max(want_s), 1.0), (size, j, i)
    for a in (t, s, dev):
        a.free()


def test_config3_gradient_16384():
    """configs[2]: 16384 x 16384 gradient at sigma 3.25 and 30.25, 1e-4 tolerance."""
    n = 16384
    dev = d.synth_dem(n, n, seed=2)
    blk = d.Block(dev)
    outs = [d.DeviceArray(n, n) for _ in range(4)]
    for sigma in (3.25, 30.25):
        R = int(4 * sigma + 0.5) + 1
        blk.gradient(sigma, [30.0], [-30.0], dx=outs[0], dy=outs[1], slope=outs[2], aspect=outs[3])
        d.sync()
        for (j, i) in windows(n, n, 128, 2, int(sigma), R):
            j0, j1, i0, i1 = max(0, j - R), min(n, j + 128 + R), max(0, i - R), min(n, i + 128 + R)
            sub = dev.to_host(j0, j1 - j0)[:, i0:i1]
            res = {"x": np.full(i1 - i0, 30.0), "y": np.full(j1 - j0, -30.0)}
            exact = orc.gradient_exact(sub, sigma, res)
            a, b = j - j0, i - i0
            interior = j0 > 0 and i0 > 0 and j1 < n and i1 < n
            for k, nm in enumerate(("dx", "dy", "slope")):
                got = outs[k].to_host(j, 128)[:, i:i + 128]
                want = exact[k][a:a + 128, b:b + 128]
                if interior or (j0 == 0 and i0 == 0):
                    sl = (slice(0, 128 - (0 if interior else R)), slice(0, 128 - (0 if interior else R)))
                    scale = max(np.max(np.abs(want)), 1e-3)
                    assert np.max(np.abs(got - want)[sl]) <= 1e-4 * scale + 2e-5, (sigma, nm, j, i)
            if interior or (j0 == 0 and i0 == 0):
                m = 128 - (0 if interior else R)
                cut = [e[a:a + m, b:b + m] for e in exact]
                tol = 1e-4 * max(np.max(np.abs(cut[0])), np.max(np.abs(cut[1])), 1e-3) + 2e-5
                check_aspect(outs[3].to_host(j, 128)[:m, i:i + m], cut, tol, (sigma, j, i))
    for a in outs + [dev]:
        a.free()


def test_config4_sx_16384():
    """configs[3]: 16384 x 16384 Sx, azimuth 0, radius 500 m."""
    n = 16384
    dev = d.synth_dem(n, n, seed=3)
    blk = d.Block(dev)
    out = d.DeviceArray(n, n)
    window, dj, di, dist = d.sx_offsets(0.0, 500.0, 30.0, -30.0)
    assert window == 17 and len(np.unique(np.stack([dj, di], 1), axis=0)) == 32
    blk.sx(dj, di, dist, window, 10.0, out)
    d.sync()
    x = 2600000.0 + 30.0 * np.arange(256 + 2 * window)
    for (j, i) in [(window, window), (5000, 9000), (n - 256 - window, n - 256 - window)]:
        sub = dev.to_host(j - window, 256 + 2 * window)[:, i - window:i + 256 + window]
        y = 1200000.0 - 30.0 * np.arange(sub.shape[0])
        want = orc.sx(sub, x, y, 0.0, 500.0)[window:-window, window:-window]
        got = out.to_host(j, 256)[:, i:i + 256]
        assert np.max(np.abs(got - want)) <= 1e-4 * np.max(np.abs(want)), (j, i)
    frame = out.to_host(0, window)
    assert np.all(frame == 0)
    out.free()
    dev.free()


def test_config5_32768():
    """configs[4] minus the wire: 32768 x 32768, TPI + STD at 67 px, gradient at sigma 3.25 and 30.25, Sx
    azimuth 0 radius 500 m, computed (i) as one block and (ii) as 8 row blocks of 4096 rows, each generated
    on the device with exactly its ghost rows and run one after the other on the one GPU.  (ii) must equal
    (i) bit for bit - what the RCCL halo exchange then has to deliver is just those ghost rows - and (i)
    must agree with the exact oracle on windows that straddle the seams at rows 4096 k."""
    n, nb = 32768, 8
    rows_b = n // nb
    dev = d.synth_dem(n, n, seed=5)
    whole_blk = d.Block(dev)
    window, dj, di, dist = d.sx_offsets(0.0, 500.0, 30.0, -30.0)
    sx_up, sx_down = int(max(0, -dj.min())), int(max(0, dj.max()))

    def shards(above, below):
        for k in range(nb):
            row0 = k * rows_b
            lo, hi = max(0, row0 - above), min(n, row0 + rows_b + below)
            part = d.synth_dem(hi - lo, n, row0=lo, seed=5)  # a shard of its own, not a view of `dev`
            yield row0, d.Block(part, row0=lo, gny=n), part

    def check_blocks(name, planes, above, below, call):
        """planes: the one-block results; call(block, row0, rows, outs) fills outs for one shard."""
        outs = [d.DeviceArray(rows_b, n) for _ in planes]
        for row0, blk, part in shards(above, below):
            call(blk, row0, rows_b, outs)
            d.sync()
            for k, (o, w) in enumerate(zip(outs, planes)):
                assert np.array_equal(o.to_host(), w.to_host(row0, rows_b)), (name, k, row0)
            part.free()
        for o in outs:
            o.free()

    seams = [rows_b * k for k in (1, 4, 7)]

    # ---- TPI + STD, 67 px ----
    size, r, w = 67, 33, 160
    t, s = d.DeviceArray(n, n), d.DeviceArray(n, n)
    whole_blk.tpi_std(size, tpi=t, std=s)
    d.sync()
    up, down = halo(_lib.DESC_TPI, size)
    check_blocks("tpi_std", [t, s], up, down,
                 lambda blk, row0, rows, o: blk.tpi_std(size, tpi=o[0], std=o[1], out_row0=row0, out_rows=rows))
    for j, i in [(0, 0), (n - w, n - w)] + [(sj - w // 2, 5000 + 3 * sj) for sj in seams]:
        i = min(i, n - w)
        j0, j1, i0, i1 = max(0, j - r), min(n, j + w + r), max(0, i - r), min(n, i + w + r)
        sub = dev.to_host(j0, j1 - j0)[:, i0:i1]
        a, b = j - j0, i - i0
        want_t = orc.tpi_exact(sub, size)[a:a + w, b:b + w]
        want_s = orc.std_exact(sub, size)[a:a + w, b:b + w]
        keep = np.ones((w, w), bool)  # pixels whose disc crosses a cut that is not a DEM edge
        if j0 > 0 and a < r: keep[: r - a] = False
        if j1 < n and j1 - (j + w) < r: keep[w - (r - (j1 - (j + w))):] = False
        if i0 > 0 and b < r: keep[:, : r - b] = False
        if i1 < n and i1 - (i + w) < r: keep[:, w - (r - (i1 - (i + w))):] = False
        assert np.max(np.abs(t.to_host(j, w)[:, i:i + w] - want_t)[keep]) <= 2.5e-4, (j, i)
        assert np.max(np.abs(s.to_host(j, w)[:, i:i + w] - want_s)[keep]) <= 1e-4 * max(np.max(want_s), 1.0), (j, i)
    t.free()
    s.free()

    # ---- gradient, sigma 3.25 and 30.25 ----
    outs = [d.DeviceArray(n, n) for _ in range(4)]
    for sigma in (3.25, 30.25):
        whole_blk.gradient(sigma, [30.0], [-30.0], dx=outs[0], dy=outs[1], slope=outs[2], aspect=outs[3])
        d.sync()
        up, down = halo(_lib.DESC_GRADIENT, sigma, 1.0)
        check_blocks(f"gradient {sigma}", outs, up, down,
                     lambda blk, row0, rows, o: blk.gradient(sigma, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2],
                                                             aspect=o[3], out_row0=row0, out_rows=rows))
        R, w = int(4 * sigma + 0.5) + 1, 128
        for sj in seams[:2]:
            j, i = sj - w // 2, 9000 + sj
            j0, j1, i0, i1 = j - R, j + w + R, i - R, i + w + R
            sub = dev.to_host(j0, j1 - j0)[:, i0:i1]
            exact = orc.gradient_exact(sub, sigma, {"x": np.full(i1 - i0, 30.0), "y": np.full(j1 - j0, -30.0)})
            for k, nm in enumerate(("dx", "dy", "slope")):
                got = outs[k].to_host(j, w)[:, i:i + w]
                want = exact[k][R:R + w, R:R + w]
                assert np.max(np.abs(got - want)) <= 1e-4 * max(np.max(np.abs(want)), 1e-3) + 2e-5, (sigma, nm, sj)
            cut = [e[R:R + w, R:R + w] for e in exact]
            tol = 1e-4 * max(np.max(np.abs(cut[0])), np.max(np.abs(cut[1])), 1e-3) + 2e-5
            check_aspect(outs[3].to_host(j, w)[:, i:i + w], cut, tol, (sigma, sj))
    for o in outs:
        o.free()

    # ---- Sx, azimuth 0, radius 500 m ----
    out = d.DeviceArray(n, n)
    whole_blk.sx(dj, di, dist, window, 10.0, out)
    d.sync()
    check_blocks("sx", [out], sx_up, sx_down,
                 lambda blk, row0, rows, o: blk.sx(dj, di, dist, window, 10.0, o[0], out_row0=row0, out_rows=rows))
    x = 2600000.0 + 30.0 * np.arange(256 + 2 * window)
    for sj in seams[:2]:
        j, i = sj - 128, 20000
        sub = dev.to_host(j - window, 256 + 2 * window)[:, i - window:i + 256 + window]
        y = 1200000.0 - 30.0 * np.arange(sub.shape[0])
        want = orc.sx(sub, x, y, 0.0, 500.0)[window:-window, window:-window]
        got = out.to_host(j, 256)[:, i:i + 256]
        assert np.max(np.abs(got - want)) <= 1e-4 * np.max(np.abs(want)), sj
    out.free()
    dev.free()


WAVE_SIZES = list(range(5, 102, 2))   # every odd size has a wave-shift instantiation


@pytest.mark.parametrize("size", WAVE_SIZES + [1, 2, 3, 4, 8, 16, 66, 103, 119])
def test_every_disc_size_against_exact(size):
    """All wave-shift instantiations plus sizes that take the generic kernel."""
    from topo_descriptors_amd import topo
    for integer in (True, False):
        dem = orc.synthetic_dem(150, 200, seed=size + 100 * integer, integer=integer)
        t, s = topo.tpi_std(dem, size)
        if size == 1:  # division by n - 1 = 0: non-finite, like the reference
            assert not np.any(np.isfinite(t))
            continue
        assert np.max(np.abs(t - orc.tpi_exact(dem, size))) <= 2.5e-4, (size, integer)
        e = orc.std_exact(dem, size)
        assert np.max(np.abs(s - e)) <= 1e-4 * np.max(e), (size, integer)
        bound = tpi_alone_bound(size, integer)
        if bound == 0.0:
            assert np.array_equal(topo.tpi(dem, size), t)
        else:  # fractional elevations under a marching kernel: x in units of 2^-8 m (see tpi_alone_bound)
            alone = topo.tpi(dem, size)
            assert np.max(np.abs(alone - t)) <= bound, size
            assert np.sqrt(np.mean((alone - t) ** 2)) <= 0.25 * bound, size  # (Gaussian noise: the errors average out)
        assert np.array_equal(topo.std(dem, size), s)


@pytest.mark.parametrize("size,level", [(7, 0.0), (7, 17000.0), (17, 0.0), (17, 6500.0), (31, 0.0), (31, 2600.0), (45, -1000.0),
                                        (65, 0.0), (67, 0.0), (67, -600.0), (67, -1500.0), (67, -2900.0), (67, -4400.0)])
def test_std_tiles_at_the_dem_border(size, level):
    """The ring kernel takes the tiles whose discs reach over the DEM's edge itself - the padding's zeros are staged as samples
    of elevation 0, so the sums run over all n taps and no tap counts enter - while the window's range [0, highest sample]
    fits the 32-bit chain of squares (2 lim32), and leaves them to the general kernel otherwise.  Levels: every border tile
    fits / none does / some do; below sea level the range is [lowest, 0].  Same exact integers either way, so row blocks -
    whose border tiles are other tiles - keep the bits."""
    from topo_descriptors_amd import topo
    dem = (orc.synthetic_dem(430, 640, seed=size) + np.float32(level)).astype(np.float32)
    t, s = topo.tpi_std(dem, size)
    e = orc.std_exact(dem, size)
    assert np.max(np.abs(s - e)) <= 1e-4 * np.max(e)
    te = orc.tpi_exact(dem, size)
    assert np.max(np.abs(t - te)) <= max(2.5e-4, 1e-7 * np.max(np.abs(te)))  # (float32 outputs: TPI at the padded edge is large)
    assert np.array_equal(topo.std(dem, size), s)
    up, down = halo(_lib.DESC_TPI, size)
    nx = dem.shape[1]

    def call(blk, row0, rows):
        tt, ss = d.DeviceArray(rows, nx), d.DeviceArray(rows, nx)
        blk.tpi_std(size, tpi=tt, std=ss, out_row0=row0, out_rows=rows)
        return [tt, ss]

    for nblocks in (2, 3):
        tb, sb = run_blocks(dem, nblocks, up, down, call)
        assert np.array_equal(tb, t) and np.array_equal(sb, s), (size, level, nblocks)


def test_nodata_and_nan_tiles_do_not_wrap():
    """-9999 nodata next to terrain exceeds the exact integer range of a tile: the float chains
    take over (no silent wrap-around); NaN poisons only windows that contain it."""
    from topo_descriptors_amd import topo
    dem = orc.synthetic_dem(200, 256, seed=77)
    dem[:, :40] = -9999.0
    for size in (7, 67):
        t, s = topo.tpi_std(dem, size)
        assert np.max(np.abs(t - orc.tpi_exact(dem, size))) <= 0.05
        e = orc.std_exact(dem, size)
        assert np.max(np.abs(s - e)) <= 2e-3 * np.max(e)
    dem = orc.synthetic_dem(200, 256, seed=78)
    dem[100, 128] = np.nan
    t = topo.tpi(dem, 7)
    assert np.isnan(t[100, 128]) and np.isnan(t[98, 127])
    clean = orc.tpi_exact(np.nan_to_num(dem, nan=2000.0), 7)
    far = np.ones_like(dem, bool)
    far[:, :] = True
    far[100 - 64:100 + 64, :] = False  # rows of the tiles the NaN can reach
    assert np.max(np.abs(t[far] - clean[far])) <= 2.5e-4


def test_rccl_communicator_single_rank():
    """RCCL is linked, the unique id round-trips through ctypes the way bench.py passes it, and a
    one-rank communicator drives the sharded entry points (no neighbours: the send/recv group is
    empty, the interior/seam split and the stream hand-over still run)."""
    lib = _lib.lib()
    uid = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
    _lib.check(lib.topo_amd_comm_unique_id(uid), "comm_unique_id")
    payload = bytes(uid.raw)           # what a broadcast would deliver to the other ranks
    assert len(payload) == _lib.UNIQUE_ID_BYTES and any(payload)
    _lib.check(lib.topo_amd_comm_init(0, 1, payload), "comm_init")
    try:
        assert lib.topo_amd_comm_size() == 1 and lib.topo_amd_comm_rank() == 0
        gny, nx, size = 256, 256, 67
        dem = orc.synthetic_dem(gny, nx, seed=41)
        up, down = halo(_lib.DESC_TPI, size)
        sd = shard.ShardedDEM(shard.RowShardPlan(gny, nx, 1, 0, up, down), dem)
        t = d.DeviceArray(gny, nx)
        for _ in range(3):             # repeated steps reuse the events / streams
            sd.tpi_std(size, tpi=t)
        d.sync()
        assert np.max(np.abs(t.to_host() - orc.tpi_exact(dem, size))) <= 2.5e-4
    finally:
        _lib.check(lib.topo_amd_comm_destroy(), "comm_destroy")


@pytest.mark.parametrize("size", [121, 151, 256, 401])
def test_very_large_discs(size):
    """Discs beyond an LDS tile take the float64 global-prefix kernel (exact, any size); the
    disc may even be larger than the DEM."""
    from topo_descriptors_amd import topo
    for integer in (True, False):
        dem = orc.synthetic_dem(150, 200, seed=size, integer=integer)
        t, s = topo.tpi_std(dem, size)
        assert np.max(np.abs(t - orc.tpi_exact(dem, size))) <= 2.5e-4, (size, integer)
        e = orc.std_exact(dem, size)
        assert np.max(np.abs(s - e)) <= 1e-4 * np.max(e), (size, integer)
    # row blocks stay bit-identical on this path too
    dem = orc.synthetic_dem(300, 128, seed=size, integer=False)
    up, down = halo(_lib.DESC_TPI, size)

    def call(blk, row0, rows):
        a, b = d.DeviceArray(rows, 128), d.DeviceArray(rows, 128)
        blk.tpi_std(size, tpi=a, std=b, out_row0=row0, out_rows=rows)
        return [a, b]

    whole = run_blocks(dem, 1, up, down, call)
    parts = run_blocks(dem, 2, up, down, call)
    assert np.array_equal(parts[0], whole[0]) and np.array_equal(parts[1], whole[1])
    # a NaN in the upper block sends that block (and the whole DEM) to the float64 planes while the
    # lower block keeps the uint32 ones: the same bits wherever both results are finite (on the
    # float64 planes a NaN travels down its column's running sum, so the whole-DEM run has more NaN)
    dem[5, 7] = np.nan
    dem[9, 100] = -3.0e6
    whole = run_blocks(dem, 1, up, down, call)
    parts = run_blocks(dem, 2, up, down, call)
    for w, q in zip(whole, parts):
        both = np.isfinite(w) & np.isfinite(q)
        assert np.array_equal(w[both], q[both])
        assert both[150:].any() or size > 151  # (the widest discs reach the NaN's column from every pixel)
        assert not np.isfinite(w[~np.isfinite(q)]).any()  # the split run is NaN only where the whole one is


@pytest.mark.parametrize("size", [7, 19, 67])
@pytest.mark.parametrize("band", [(0, 700), (300, 1100), (900, 1500), (640, 700)])
def test_std_row_blocks_on_a_dem_with_a_fractional_band(size, band):
    """STD / TPI+STD on rows of whole metres next to rows with fractional elevations.  The ring kernels hand a run
    whose probed rows are all fractional to the general kernel unstaged, so which kernel computes a pixel depends
    on the run's extent, i.e. on the row block: both must give a window of whole metres the same bits (the
    general kernel picks its variance expression per pixel; a per-tile choice failed the randomised check once
    in 72 206 cases)."""
    ny, nx = 1500, 512
    dem = orc.synthetic_dem(ny, nx, seed=23, integer=True).astype(np.float32)
    dem[band[0]:band[1]] += np.float32(0.37)
    up, down = halo(_lib.DESC_STD, size)

    def call(blk, row0, rows):
        t, s = d.DeviceArray(rows, nx), d.DeviceArray(rows, nx)
        blk.tpi_std(size, tpi=t, std=s, out_row0=row0, out_rows=rows)
        return [t, s]

    whole = run_blocks(dem, 1, up, down, call)
    for nb in (3, 4):
        parts = run_blocks(dem, nb, up, down, call)
        assert np.array_equal(whole[0], parts[0]), nb
        assert np.array_equal(whole[1], parts[1]), nb
    e = orc.std_exact(dem, size)
    assert np.max(np.abs(whole[1] - e)) <= 1e-4 * max(float(np.max(e)), 1.0)


@pytest.mark.parametrize("layout", ["fractional_first", "fractional_band", "fractional_rows"])
def test_tpi_fast_and_deferred_tiles_on_a_mixed_dem(layout):
    """TPI alone runs as a fast build plus a deferred pass over the tiles that need the fractional /
    float paths (csrc/disc_wave_impl.hpp).  4096 x 4096 gives ~1500 tiles at 67 px, 5-6 per
    persistent block, so blocks defer, give up after four deferrals in a row (fractional_first: the
    integer tiles behind them go to the general build unstaged) or go back to the fast path after a
    deferred stretch (fractional_band).  Which build takes a tile must not show: exact-oracle
    agreement on windows across the region edges, and bit-identity with the same DEM computed in
    three row blocks (another tile list, so another tile -> build assignment).
    fractional_rows: a fractional band of rows; just below it, and at the bottom of the DEM in the
    other layouts, a marched tile adds only whole-metre (or out-of-DEM) rows to a window whose
    carried rows are fractional - the classification has to remember them."""
    n, size, r = 4096, 67, 33
    dev = d.synth_dem(n, n, seed=11)
    dem = dev.to_host()
    dev.free()
    c0, c1 = (0, 2800) if layout == "fractional_first" else (1000, 2500)
    if layout == "fractional_rows":
        dem[c0:c1, :] += np.float32(0.37)
        assert np.any(dem[c0:c1] != np.trunc(dem[c0:c1])) and np.all(dem[c1:] == np.trunc(dem[c1:]))
    else:
        dem[:, c0:c1] += np.float32(0.37)
        assert np.any(dem[:, c0:c1] != np.trunc(dem[:, c0:c1])) and np.all(dem[:, c1:] == np.trunc(dem[:, c1:]))
    up, down = halo(_lib.DESC_TPI, size)

    def call(blk, row0, rows):
        t = d.DeviceArray(rows, n)
        blk.tpi_std(size, tpi=t, out_row0=row0, out_rows=rows)
        return [t]

    whole = run_blocks(dem, 1, up, down, call)[0]
    parts = run_blocks(dem, 3, up, down, call)[0]
    assert np.array_equal(whole, parts), layout

    w = 192
    corners = [(0, max(c0 - w // 2, 0)), (1500, c1 - w // 2), (n - w, c1 - w // 2), (2000, c1 + 600),
               (700, (c0 + c1) // 2), (n - w, n - w)]
    if layout == "fractional_rows":  # the band's upper and lower edges, and the DEM's bottom edge
        corners = [(c0 - w // 2, 300), (c1 - w // 2, 1900), (c1 - 20, 3500), (c1 + 30, 100), (n - w, 2000), (n - w, 0)]
    for (j, i) in corners:
        i = min(max(i, 0), n - w)
        j0, j1, i0, i1 = max(0, j - r), min(n, j + w + r), max(0, i - r), min(n, i + w + r)
        a, b = j - j0, i - i0
        want = orc.tpi_exact(dem[j0:j1, i0:i1], size)[a:a + w, b:b + w]
        got = whole[j:j + w, i:i + w]
        keep = np.ones((w, w), bool)  # pixels whose disc crosses a cut that is not a DEM edge
        if j0 > 0 and a < r: keep[: r - a] = False
        if j1 < n and j1 - (j + w) < r: keep[w - (r - (j1 - (j + w))):] = False
        if i0 > 0 and b < r: keep[:, : r - b] = False
        if i1 < n and i1 - (i + w) < r: keep[:, w - (r - (i1 - (i + w))):] = False
        assert keep.any()
        # (+ 0.37 m everywhere is the worst case of the scaled route: every sample is off the 2^-8 m grid by the same 1.1 mm)
        assert np.max(np.abs(got - want)[keep]) <= 2.5e-4 + tpi_alone_bound(size), (layout, j, i)


_WRAP_CHILD = r"""
import numpy as np
from oracle import topo_oracle as orc
from topo_descriptors_amd import topo, _lib
assert _lib.lib().topo_amd_cu_count() == 8, "TOPO_AMD_CU_LIMIT did not take effect: the runs would be too short to wrap"
ny, nx, size, r = 16384, 768, 67, 33
rng = np.random.default_rng(5)
dem = (250000.0 + rng.integers(-60, 61, size=(ny, nx))).astype(np.float32)
dem[:, 400:] -= 500000.0          # a strip of large negative values: the prefix wraps downwards there
got = topo.tpi(dem, size)
worst = 0.0
for j in (0, 4000, 8500, 8700, 10200, 12000, ny - 160):
    for i in (0, 300, nx - 160):
        j0, j1, i0, i1 = max(0, j - r), min(ny, j + 160 + r), max(0, i - r), min(nx, i + 160 + r)
        want = orc.tpi_exact(dem[j0:j1, i0:i1], size)[j - j0:j - j0 + 160, i - i0:i - i0 + 160]
        # 2.5e-4 m where TPI is noise-sized; one float32 ulp where it is ~5e5 m (across the sign change)
        tol = np.maximum(2.5e-4, np.spacing(np.abs(want).astype(np.float32)).astype(np.float64))
        worst = max(worst, float(np.max(np.abs(got[j:j + 160, i:i + 160] - want) / tol)))
print("WORST", worst)
"""


def test_marching_prefix_wraps_harmlessly():
    """The marching TPI kernel keeps a running uint32 prefix down a column strip and relies on
    wrap-around being harmless.  With the grid limited to 8 blocks (TOPO_AMD_CU_LIMIT, read at
    init, hence a child process) a 16384-row DEM of +-250000 m gives runs of ~170 tiles: the
    prefix passes 2^31 after ~8600 rows (and -2^31 in the negative strip).  Exact-oracle agreement
    on windows before, at and after the wrap."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, TOPO_AMD_CU_LIMIT="8")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _WRAP_CHILD], cwd=root, env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    worst = float(out.stdout.strip().splitlines()[-1].split()[1])  # max |gpu - exact| / tolerance
    assert worst <= 1.0, worst


@pytest.mark.gpu
@pytest.mark.parametrize("nx", [257, 258, 259])
@pytest.mark.parametrize("size", [3, 17, 67])
def test_widths_that_are_not_multiples_of_four(nx, size):
    """Rows that are not 16-byte aligned go through a re-pitched copy with zero columns on the right:
    the result equals, bit for bit, the valid columns of the same DEM widened with zeros."""
    ny = 200
    rng = np.random.default_rng(nx * 100 + size)
    dem = np.rint(orc.synthetic_dem(ny, nx, seed=size)).astype(np.float32)
    if nx == 258:
        dem += rng.uniform(0, 1, dem.shape).astype(np.float32)  # fractional tiles too
    dev = d.DeviceArray.from_host(dem)
    t, s = d.DeviceArray(ny, nx), d.DeviceArray(ny, nx)
    d.Block(dev).tpi_std(size, tpi=t, std=s)
    d.sync()
    got_t, got_s = t.to_host(), s.to_host()
    nxp = (nx + 3) // 4 * 4
    wide = np.zeros((ny, nxp), dtype=np.float32)
    wide[:, :nx] = dem
    devw = d.DeviceArray.from_host(wide)
    tw, sw = d.DeviceArray(ny, nxp), d.DeviceArray(ny, nxp)
    d.Block(devw).tpi_std(size, tpi=tw, std=sw)
    d.sync()
    assert np.array_equal(got_t, tw.to_host()[:, :nx])
    assert np.array_equal(got_s, sw.to_host()[:, :nx])
    assert np.max(np.abs(got_t - orc.tpi_exact(dem, size))) <= 2.5e-4
    e = orc.std_exact(dem, size)
    assert np.max(np.abs(got_s - e)) <= 1e-4 * max(np.max(e), 1.0)
    for a in (dev, t, s, devw, tw, sw):
        a.free()


def test_gradient_row_chunks_bit_identical(tmp_path):
    """From 8192 rows on, the matrix-core gradient goes in row chunks with the epilogue of a chunk on a second
    stream beside the smooth of the next one (csrc/gauss.hip launch_gradient).  Chunks are row blocks: the bits
    must not depend on the cut.  The threshold is read once per process, hence two child processes."""
    import subprocess
    import sys
    script = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from oracle import topo_oracle as orc\n"
        "import topo_descriptors_amd.topo as topo\n"
        "dem = orc.synthetic_dem(700, 512, seed=5)\n"
        "np.save(sys.argv[1], np.stack(topo.gradient(dem, 9.0, {'x': 50.0, 'y': -50.0}) + topo.gradient(dem, 11.0, {'x': 50.0, 'y': -50.0})))\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for name, rows in (("chunked", "64"), ("whole", "100000000")):
        out = str(tmp_path / (name + ".npy"))
        env = dict(os.environ, TOPO_AMD_GRAD_CHUNK_MIN_ROWS=rows)
        subprocess.check_call([sys.executable, "-c", script, out], env=env)
        outs.append(np.load(out))
    assert np.array_equal(outs[0], outs[1], equal_nan=True)
    dem = orc.synthetic_dem(700, 512, seed=5)
    exact = orc.gradient_exact(dem, 9.0, {"x": 50.0, "y": -50.0})
    assert np.max(np.abs(outs[0][2] - exact[2])) <= 1e-4 * max(1.0, float(np.max(np.abs(exact[2]))))





def test_requests_taller_than_one_launch():
    """Several kernels launch one block row per DEM row, and a launch covers 65 535 of them: the launch_* entry points
    cut taller requests into row blocks themselves (kMaxLaunchRows), and row blocks give the single block's bits.
    70 016 rows x 64 columns in one call against two calls of 35 008 rows: Sobel, the fused and the two-pass
    Gaussian, the gradient with and without row chunks, a generic (even) disc size and a ring-kernel size."""
    ny, nx, half = 70016, 64, 35008
    dem = d.DeviceArray(ny, nx)
    d.synth_dem(40000, nx, row0=0, seed=5, out=dem, out_row=0)
    d.synth_dem(ny - 40000, nx, row0=40000, seed=5, out=dem, out_row=40000)
    blk = d.Block(dem)

    def both_ways(nplanes, call):
        whole = [d.DeviceArray(ny, nx) for _ in range(nplanes)]
        call(whole, None, None)
        d.sync()
        w = [a.to_host() for a in whole]
        for o0, on in ((0, half), (half, ny - half)):
            part = [d.DeviceArray(on, nx) for _ in range(nplanes)]
            call(part, o0, on)
            d.sync()
            for k in range(nplanes):
                assert np.array_equal(part[k].to_host(), w[k][o0:o0 + on], equal_nan=True), (k, o0)
            for a in part:
                a.free()
        for a in whole:
            a.free()
        return w

    for sigma in (0.75, 2.0, 9.0):
        both_ways(4, lambda o, o0, on: blk.gradient(sigma, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3],
                                                    out_row0=o0, out_rows=on))
    for sigma in (2.0, 9.0):
        g = both_ways(1, lambda o, o0, on: blk.gaussian(sigma, sigma, o[0], out_row0=o0, out_rows=on))[0]
        host = dem.to_host(69000, 1016)
        from scipy import ndimage
        ref = ndimage.gaussian_filter(host, sigma, mode="reflect")
        assert np.max(np.abs(g[69000 + 40:] - ref[40:])) <= 1e-3  # the last rows, beyond the first launch's reach
    for size in (6, 7):
        both_ways(2, lambda o, o0, on: blk.tpi_std(size, tpi=o[0], std=o[1], out_row0=o0, out_rows=on))
    dem.free()


@pytest.mark.parametrize("azimuth", [45.0, 135.0, 225.0, 315.0, 30.0, 0.0, 90.0, 180.0])
def test_sx_diagonal_chains(azimuth):
    """(Azimuths 0, 90, 180: the paired chains of a sector that points along an axis - the larger of two samples at
    equal distance decides - against the same three references.)
    Sectors off the axes at a radius where the launcher scans along a diagonal (sx_kernel<.., DIAG = +-1>, from ~100
    comparisons per pixel): the oracle, the axis-aligned scan of the same sector (TOPO_AMD_SX_DIAG=0 would give the
    same bits: a maximum does not care about the order; here checked against the fan kernel, which never scans
    diagonally), row blocks, and a width where the leaning slabs meet both DEM edges."""
    gny, nx = 330, 300
    dem = orc.synthetic_dem(gny, nx, seed=23)
    window, dj, di, dist = d.sx_offsets(azimuth, 1500.0, 30.0, -30.0)
    up, down = halo(_lib.DESC_SX, max(0, -dj.min()), max(0, dj.max()))

    def call(blk, row0, rows):
        out = d.DeviceArray(rows, nx)
        blk.sx(dj, di, dist, window, 10.0, out, out_row0=row0, out_rows=rows)
        return [out]

    whole = run_blocks(dem, 1, up, down, call)[0]
    for nb in (2, 3):
        assert np.array_equal(run_blocks(dem, nb, up, down, call)[0], whole, equal_nan=True), (azimuth, nb)
    x = 2600000.0 + 30.0 * np.arange(nx)
    y = 1200000.0 - 30.0 * np.arange(gny)
    want = orc.sx(dem, x, y, azimuth, 1500.0)
    assert np.max(np.abs(whole - want)) <= 1e-4 * np.max(np.abs(want))
    # the fan kernel on the same sector (and a neighbour, so that it is a fan): plane for plane the same bits
    dev = d.DeviceArray.from_host(dem)
    outs = [d.DeviceArray(gny, nx) for _ in range(2)]
    d.Block(dev).sx_multi([(window, dj, di, dist), d.sx_offsets(azimuth + 5.0, 1500.0, 30.0, -30.0)], 10.0, outs)
    d.sync()
    assert np.array_equal(outs[0].to_host(), whole, equal_nan=True)
    for a in outs:
        a.free()
    dev.free()


# ---- fractional elevations under the marching TPI kernels: the scaled one-chain route (task 2) -----------
_SCALED_CHILD = r"""
import sys, zlib, json
import numpy as np
sys.path.insert(0, %r)
from oracle import topo_oracle as orc
from topo_descriptors_amd import device as d
n, size = 2048, 67
rng = np.random.default_rng(3)
dem = orc.synthetic_dem(n, n, seed=4) + rng.uniform(0, 1, (n, n)).astype(np.float32)   # every tile fractional
dem[300:420, 900:1300] = np.rint(dem[300:420, 900:1300])                                   # ... but one patch of whole metres
dev = d.DeviceArray.from_host(dem.astype(np.float32))
out = d.DeviceArray(n, n)
blk = d.Block(dev)
crcs = []
for call in range(4):   # first call: whole-metre kernel, then the scaled one; later calls: the library remembers the DEM
    blk.tpi_std(size, tpi=out)
    d.sync()
    crcs.append(zlib.crc32(out.to_host().tobytes()))
np.save(sys.argv[1], out.to_host())
print(json.dumps({"crcs": crcs}))
"""


def test_scaled_fraction_route_against_the_exact_one(tmp_path):
    """TPI 67 px on a DEM whose elevations are fractional (uniform fractional parts): the scaled route (x in units of
    2^-8 m, one chain) against the exact two-pass route (TOPO_AMD_TPI_FRACTION_EXACT=1) - within 2^-9 m everywhere,
    far closer in the mean; a patch of whole metres keeps the exact bits where the discs see nothing else; and the
    call that starts with the take-all scaled kernel (what the library does once it remembers the DEM as fractional)
    returns the bits of the call that started with the whole-metre kernel."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    planes = {}
    for name, env in (("scaled", {}), ("exact", {"TOPO_AMD_TPI_FRACTION_EXACT": "1"})):
        path = str(tmp_path / (name + ".npy"))
        out = subprocess.run([sys.executable, "-c", _SCALED_CHILD % root, path], cwd=root, env=dict(os.environ, **env),
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (name, out.stderr[-3000:])
        crcs = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])["crcs"]
        assert len(set(crcs)) == 1, (name, crcs)  # whichever kernel goes first
        planes[name] = np.load(path)
    err = np.abs(planes["scaled"].astype(np.float64) - planes["exact"])
    assert err.max() <= 2.0 ** -9
    assert np.sqrt(np.mean(err ** 2)) <= 1e-4   # (uniform fractional parts: 2^-8 / sqrt(12 n) = 1.9e-5 m)
    inner = (slice(300 + 33, 420 - 33), slice(900 + 33, 1300 - 33))
    assert np.array_equal(planes["scaled"][inner], planes["exact"][inner])


# ---- the 512-column form of the small discs' STD kernel (round 5): the same bits as the 256-column form ------------------------
_WIDE_CHILD = r"""
import sys, zlib, json
sys.path.insert(0, %r)
from topo_descriptors_amd import device as d
out = {}
for ny, nx in ((8192, 16384), (2000, 1500)):   # STD alone takes the wide form on big rasters only; TPI + STD on both
    dem = d.synth_dem(ny, nx, seed=5)
    blk = d.Block(dem)
    t, s = d.DeviceArray(ny, nx), d.DeviceArray(ny, nx)
    for size in (5, 7, 9, 13):
        blk.tpi_std(size, std=s)
        d.sync()
        a = zlib.crc32(s.to_host().tobytes())
        blk.tpi_std(size, tpi=t, std=s)
        d.sync()
        out["%%d_%%d_%%d" %% (ny, nx, size)] = [a, zlib.crc32(t.to_host().tobytes()), zlib.crc32(s.to_host().tobytes())]
    for x in (t, s, dem):
        x.free()
print(json.dumps(out))
"""


def test_std_512_column_strips_keep_the_bits():
    """STD / TPI + STD at 5 and 7 px (TPI + STD up to 13 px) on rasters of whole metres take 512-column strips
    (std_ring_spec_kernel<., ., false, 8>); TOPO_AMD_STD_SPEC_WIDE=0 keeps the 256-column form.  Exact integer sums and one
    finalisation: identical planes, on a raster large enough for STD alone to take the wide form and on a small one."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for wide in ("1", "0"):
        out = subprocess.run([sys.executable, "-c", _WIDE_CHILD % root], cwd=root, env=dict(os.environ, TOPO_AMD_STD_SPEC_WIDE=wide),
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        got[wide] = json.loads([line for line in out.stdout.splitlines() if line.startswith("{")][-1])
    assert got["1"] == got["0"]
