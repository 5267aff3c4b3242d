"""Several disc sizes from one pass over the DEM (the multi-scale half): ``topo_amd_tpi_multi_dev``
evaluates pairs of small sizes (5 ... 11 px) with the two-disc ring kernel (csrc/disc_pair.hip).  Every plane must
have the bits of the single-size call, which the parity tests check against the reference's ``topo.tpi``
(topo.py:145-181, scale loop :132-141)."""
import numpy as np
import pytest

from oracle import topo_oracle as orc

pytestmark = pytest.mark.gpu

from topo_descriptors_amd import _lib, device as d, shard  # noqa: E402


def single(blk, size, rows, nx, **kw):
    out = d.DeviceArray(rows, nx)
    blk.tpi_std(size, tpi=out, **kw)
    d.sync()
    h = out.to_host()
    out.free()
    return h


def multi(blk, sizes, rows, nx, **kw):
    outs = [d.DeviceArray(rows, nx) for _ in sizes]
    blk.tpi_multi(sizes, outs, **kw)
    d.sync()
    hs = [o.to_host() for o in outs]
    for o in outs:
        o.free()
    return hs


@pytest.mark.parametrize("sizes", [(5, 7), (7, 5), (5, 9), (5, 11), (7, 9), (11, 7), (9, 11), (5, 7, 9, 11), (11, 9, 7),
                                   (7, 67, 11, 17), (7, 7), (6, 7, 9)])
@pytest.mark.parametrize("kind", ["int", "frac", "mixed"])
def test_planes_have_the_bits_of_the_single_calls(sizes, kind):
    gny, nx = 700, 1024
    dem = orc.synthetic_dem(gny, nx, seed=sum(sizes), integer=kind == "int")
    if kind == "mixed":  # whole metres with a band of fractional rows and a NaN: tiles that go to the general kernel
        dem = np.rint(dem)
        dem[300:340] += 0.25
        dem[500, 600] = np.nan
    dev = d.DeviceArray.from_host(dem)
    blk = d.Block(dev)
    got = multi(blk, sizes, gny, nx)
    for size, plane in zip(sizes, got):
        assert np.array_equal(plane, single(blk, size, gny, nx), equal_nan=True), (sizes, size, kind)
    if kind == "int":
        assert np.max(np.abs(got[0] - orc.tpi_exact(dem, sizes[0]))) <= 2.5e-4
    dev.free()


def test_row_blocks_and_odd_widths():
    """Row blocks with ghost rows give the single block's bits; a width that is not a multiple of 4 takes the
    single-size launches (the two-disc kernel needs 16-byte rows) and still answers."""
    gny, nx = 500, 512
    dem = orc.synthetic_dem(gny, nx, seed=77)
    sizes = (7, 11)
    dev = d.DeviceArray.from_host(dem)
    whole = multi(d.Block(dev), sizes, gny, nx)
    dev.free()
    up, down = shard.halo_rows(_lib.DESC_TPI, max(sizes))
    for nb in (2, 3):
        parts = [[] for _ in sizes]
        for row0, rows in shard.split_rows(gny, nb):
            lo, hi = max(0, row0 - up), min(gny, row0 + rows + down)
            part = d.DeviceArray.from_host(dem[lo:hi])
            got = multi(d.Block(part, row0=lo, gny=gny), sizes, rows, nx, out_row0=row0, out_rows=rows)
            for p, g in zip(parts, got):
                p.append(g)
            part.free()
        for w, p in zip(whole, parts):
            assert np.array_equal(w, np.concatenate(p, axis=0)), nb
    odd = orc.synthetic_dem(300, 250, seed=5)
    dev = d.DeviceArray.from_host(odd)
    blk = d.Block(dev)
    for size, plane in zip(sizes, multi(blk, sizes, 300, 250)):
        assert np.array_equal(plane, single(blk, size, 300, 250))
        assert np.max(np.abs(plane - orc.tpi_exact(odd, size))) <= 2.5e-4
    dev.free()


def test_pair_kernel_can_be_switched_off_and_refuses_nothing(monkeypatch):
    """A size list with a missing plane is an argument error, not a crash."""
    dem = orc.synthetic_dem(64, 64, seed=1)
    dev = d.DeviceArray.from_host(dem)
    with pytest.raises(ValueError):
        d.Block(dev).tpi_multi([5, 7], [d.DeviceArray(64, 64)])
    dev.free()
