"""Sx for several azimuth sectors in one pass (C ABI topo_amd_sx_multi_*): every plane has the bits
of the single-azimuth entry point, which tests/test_gpu_parity.py and test_gpu_blocks.py hold against
the oracle; one case is checked against the oracle here too."""
import numpy as np
import pytest

from oracle import topo_oracle as orc

pytestmark = pytest.mark.gpu

from topo_descriptors_amd import _lib, device as d, shard, topo  # noqa: E402


class FakeVar:
    def __init__(self, values, dims):
        self.values, self.dims = values, dims


class FakeDataset:
    def __init__(self, dem, x, y, crs="epsg:2056"):
        self._v = {"dem": FakeVar(dem, ("y", "x")), "x": FakeVar(x, ("x",)), "y": FakeVar(y, ("y",))}
        self.attrs = {"crs": crs}

    def __getitem__(self, k):
        return self._v[k]

    def __iter__(self):
        return iter(["dem"])


def single_planes(blk, sectors, rows, nx, height=10.0):
    out = d.DeviceArray(rows, nx)
    planes = []
    for window, dj, di, dist in sectors:
        blk.sx(dj, di, dist, window, height, out)
        d.sync()
        planes.append(out.to_host())
    out.free()
    return planes


def multi_planes(blk, sectors, rows, nx, height=10.0, **kw):
    outs = [d.DeviceArray(rows, nx) for _ in sectors]
    blk.sx_multi(sectors, height, outs, **kw)
    d.sync()
    planes = [o.to_host() for o in outs]
    for o in outs:
        o.free()
    return planes


@pytest.mark.parametrize("azimuths, radius", [
    ([0.0], 300.0),                                       # a group of one: the single-azimuth path
    ([0.0, 5.0], 300.0),
    ([350.0, 355.0, 0.0, 5.0, 10.0], 500.0),              # overlapping sectors across north
    (list(np.arange(0.0, 90.0, 5.0)), 500.0),             # 18 sectors: several launches of 8
    (list(np.arange(0.0, 360.0, 45.0)), 1000.0),          # disjoint sectors, union tile too large
    ([90.0, 90.0, 91.0], 300.0),                          # identical sectors
    ([10.0, 200.0, 15.0], 2000.0),                        # far apart, large window
])
def test_every_plane_has_the_bits_of_the_single_azimuth_call(azimuths, radius):
    ny, nx = 330, 410
    rng = np.random.default_rng(int(radius) + len(azimuths))
    dem = (orc.synthetic_dem(ny, nx, seed=3) + rng.uniform(-0.5, 0.5, (ny, nx))).astype(np.float32)
    dev = d.DeviceArray.from_host(dem)
    blk = d.Block(dev)
    sectors = [d.sx_offsets(a, radius, 30.0, -30.0) for a in azimuths]
    want = single_planes(blk, sectors, ny, nx)
    got = multi_planes(blk, sectors, ny, nx)
    for a, w, g in zip(azimuths, want, got):
        assert np.array_equal(w, g, equal_nan=True), (a, radius)
    dev.free()


def test_against_the_oracle_with_radius_min_and_anisotropic_grid():
    ny, nx = 240, 300
    dem = orc.synthetic_dem(ny, nx, seed=21)
    x = 2600000.0 + 25.0 * np.arange(nx)
    y = 1200000.0 - 40.0 * np.arange(ny)
    azimuths = [40.0, 45.0, 50.0, 55.0]
    sectors = [d.sx_offsets(a, 400.0, 25.0, -40.0, radius_min=120.0) for a in azimuths]
    dev = d.DeviceArray.from_host(dem)
    got = multi_planes(d.Block(dev), sectors, ny, nx, height=2.0)
    for a, g in zip(azimuths, got):
        want = orc.sx(dem, x, y, a, 400.0, height=2.0, radius_min=120.0)
        assert np.max(np.abs(g - want)) <= 1e-4 * np.max(np.abs(want)), a
    dev.free()


def test_row_blocks_bit_identical():
    gny, nx = 300, 260
    dem = orc.synthetic_dem(gny, nx, seed=13)
    sectors = [d.sx_offsets(a, 500.0, 30.0, -30.0) for a in (120.0, 125.0, 130.0, 135.0, 140.0)]
    up, down = shard.sx_multi_halo(sectors)
    dev = d.DeviceArray.from_host(dem)
    whole = multi_planes(d.Block(dev), sectors, gny, nx)
    dev.free()
    for nb in (2, 3):
        pieces = [[] for _ in sectors]
        for row0, rows in shard.split_rows(gny, nb):
            lo, hi = max(0, row0 - up), min(gny, row0 + rows + down)
            part = d.DeviceArray.from_host(dem[lo:hi])
            got = multi_planes(d.Block(part, row0=lo, gny=gny), sectors, rows, nx, out_row0=row0, out_rows=rows)
            for p, g in zip(pieces, got):
                p.append(g)
            part.free()
        for w, p in zip(whole, pieces):
            assert np.array_equal(w, np.concatenate(p, axis=0), equal_nan=True), nb


def test_single_rank_shard_entry_point():
    gny, nx = 280, 256
    dem = orc.synthetic_dem(gny, nx, seed=5)
    sectors = [d.sx_offsets(a, 300.0, 30.0, -30.0) for a in (0.0, 5.0, 10.0)]
    up, down = shard.sx_multi_halo(sectors)
    sd = shard.ShardedDEM(shard.RowShardPlan(gny, nx, 1, 0, up, down), dem)
    outs = [d.DeviceArray(gny, nx) for _ in sectors]
    sd.sx_multi(sectors, 10.0, outs)
    d.sync()
    dev = d.DeviceArray.from_host(dem)
    want = single_planes(d.Block(dev), sectors, gny, nx)
    for w, o in zip(want, outs):
        assert np.array_equal(w, o.to_host(), equal_nan=True)
        o.free()
    dev.free()


def test_sector_without_usable_ray_pixel_is_reported_and_the_others_still_run():
    ny, nx = 120, 140
    dem = orc.synthetic_dem(ny, nx, seed=2)
    good = d.sx_offsets(30.0, 300.0, 30.0, -30.0)
    window, dj, di, dist = d.sx_offsets(35.0, 300.0, 30.0, -30.0)
    empty = (window, dj, di, np.full(dist.shape, np.nan))
    dev = d.DeviceArray.from_host(dem)
    blk = d.Block(dev)
    outs = [d.DeviceArray(ny, nx) for _ in range(3)]
    with pytest.raises(_lib.TopoAmdError, match="no usable ray pixel"):
        blk.sx_multi([good, empty, good], 10.0, outs)
    d.sync()
    want = single_planes(blk, [good], ny, nx)[0]
    assert np.array_equal(outs[0].to_host(), want) and np.array_equal(outs[2].to_host(), want)
    hole = outs[1].to_host()
    assert (hole == 0).all()  # as topo_amd_sx_dev leaves it; topo.sx / topo.sx_multi fill NaN on the host
    for o in outs:
        o.free()
    dev.free()


def test_topo_sx_multi_matches_topo_sx():
    ny, nx = 150, 170
    dem = orc.synthetic_dem(ny, nx, seed=8)
    ds = FakeDataset(dem, 2600000.0 + 30.0 * np.arange(nx), 1200000.0 - 30.0 * np.arange(ny))
    azimuths = [0.0, 5.0, 10.0, 200.0]
    got = topo.sx_multi(ds, azimuths, 300.0, radius_min=60.0)
    assert len(got) == 4
    for a, g in zip(azimuths, got):
        assert np.array_equal(g, topo.sx(ds, a, 300.0, radius_min=60.0), equal_nan=True), a
    # everything masked by radius_min: NaN inside the frame, like the single call
    far = topo.sx_multi(ds, [0.0, 5.0], 300.0, radius_min=1000.0)
    one = topo.sx(ds, 0.0, 300.0, radius_min=1000.0)
    assert np.array_equal(far[0], one, equal_nan=True)
    assert topo.sx_multi(ds, [], 300.0) == []
    with pytest.raises(TypeError):
        topo.sx_multi(dem, [0.0], 300.0)
