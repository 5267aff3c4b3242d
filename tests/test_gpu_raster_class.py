"""The raster class is a property of a raster's MEMORY, not of a thread or of the process (ADVICE r05
medium 2): declarations are keyed by the device rows they were made for and the raster's shape, die with the data, and the
``topo_amd_shard_*`` calls make them themselves.  SURVEY 8e: row blocks and shards are bit-identical to the single block."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import topo_oracle as orc

pytestmark = pytest.mark.gpu

from topo_descriptors_amd import _lib, device as d, shard, topo  # noqa: E402


def fractional_window(seed, ny=192, nx=482):
    """The generator of tools/fuzz_repro_r05.py: a window of a few hundred metres of relief with fractional elevations - the
    whole raster's scaled TPI route takes a finer unit (2^-11 ... 2^-13 m) than an ordinary DEM's 2^-8."""
    rng = np.random.default_rng(seed)
    return (orc.synthetic_dem(ny, nx, seed=seed, row0=int(rng.integers(0, 5000)), col0=int(rng.integers(0, 5000))) +
            rng.random((ny, nx))).astype(np.float32)


def blocks_of(dem, nb, up, down):
    """[(Block with its ghost rows, first owned row, owned rows)] - all resident at once."""
    gny = dem.shape[0]
    out = []
    for row0, rows in shard.split_rows(gny, nb):
        lo, hi = max(0, row0 - up), min(gny, row0 + rows + down)
        dev = d.DeviceArray.from_host(dem[lo:hi])
        out.append((d.Block(dev, row0=lo, gny=gny), row0, rows))
    return out


def declare(blocks):
    scan = d.RasterScan()
    for blk, row0, rows in blocks:
        scan.add(blk, own_row0=row0, own_rows=rows)
    scan.declare(*[b for b, _, _ in blocks])


def tpi_of(blocks, size, nx):
    parts = []
    for blk, row0, rows in blocks:
        out = d.DeviceArray(rows, nx)
        blk.tpi_std(size, tpi=out, out_row0=row0, out_rows=rows)
        d.sync()
        parts.append(out.to_host())
        out.free()
    return np.concatenate(parts)


def gauss_of(blocks, sigma, nx):
    parts = []
    for blk, row0, rows in blocks:
        out = d.DeviceArray(rows, nx)
        blk.gaussian(sigma, sigma, out, out_row0=row0, out_rows=rows)
        d.sync()
        parts.append(out.to_host())
        out.free()
    return np.concatenate(parts)


def free_blocks(blocks):
    for blk, _, _ in blocks:
        blk.data.free()


def sensitive_seed(size, nb):
    """A seed of the generator on which UNDECLARED row blocks differ from the whole raster (so that the tests below would
    notice a class that is not found)."""
    up, down = shard.halo_rows(_lib.DESC_TPI, size)
    for seed in range(40):
        dem = fractional_window(seed)
        whole = topo.tpi(dem, size)
        blocks = blocks_of(dem, nb, up, down)
        got = tpi_of(blocks, size, dem.shape[1])
        free_blocks(blocks)
        if not np.array_equal(got, whole):
            return seed, dem, whole
    pytest.fail("no seed of the fuzz generator tells undeclared blocks from the whole raster any more")


def test_two_rasters_in_one_process_keep_their_own_class():
    """A raster in millimetres (Gaussian on the vector-ALU kernels) and a fractional window of small relief (finer TPI unit),
    both held as row blocks at the same time, driven alternately from one thread: each block finds the class declared for ITS
    memory."""
    size, sigma, nb = 19, 3.25, 3
    _, frac, frac_tpi = sensitive_seed(size, nb)
    mm = (orc.synthetic_dem(300, 512, seed=6, integer=False) * 1000.0).astype(np.float32)
    mm_gauss = topo.dem(mm, sigma)
    frac_gauss = topo.dem(frac, sigma)
    up_t, down_t = shard.halo_rows(_lib.DESC_TPI, size)
    up_g, down_g = shard.halo_rows(_lib.DESC_GAUSS, sigma)
    up, down = max(up_t, up_g), max(down_t, down_g)
    b_frac = blocks_of(frac, nb, up, down)
    b_mm = blocks_of(mm, 2, up, down)
    declare(b_frac)
    declare(b_mm)
    for _ in range(2):  # interleaved, twice
        parts_f, parts_m, parts_g = [], [], []
        for k in range(max(len(b_frac), len(b_mm))):
            if k < len(b_frac):
                parts_f.append(tpi_of([b_frac[k]], size, frac.shape[1]))
                parts_g.append(gauss_of([b_frac[k]], sigma, frac.shape[1]))
            if k < len(b_mm):
                parts_m.append(gauss_of([b_mm[k]], sigma, mm.shape[1]))
        assert np.array_equal(np.concatenate(parts_f), frac_tpi)
        assert np.array_equal(np.concatenate(parts_g), frac_gauss)
        assert np.array_equal(np.concatenate(parts_m), mm_gauss)
    # what each block would take
    dec, large, lo, hi, share = d.raster_class(b_mm[0][0])
    assert dec and large and share > 0.5
    dec, large, lo, hi, share = d.raster_class(b_frac[1][0])
    assert dec and not large and hi - lo < 4096.0 and share > 0.9
    free_blocks(b_frac)
    free_blocks(b_mm)


def test_a_declaration_lives_as_long_as_the_data():
    dem = fractional_window(3)
    gny, nx = dem.shape
    dev = d.DeviceArray.from_host(dem[:100])
    blk = d.Block(dev, row0=0, gny=gny)
    other = d.DeviceArray.from_host(dem[100:])
    blk2 = d.Block(other, row0=100, gny=gny)
    assert d.raster_class(blk)[0] is False
    d.RasterScan().add(blk).add(blk2).declare(blk, blk2)
    assert d.raster_class(blk)[0] and d.raster_class(blk2)[0]
    # a view into declared rows finds the class; the same memory as rows of a raster of ANOTHER shape does not
    assert d.raster_class(d.Block(dev, row0=10, gny=gny, first_buffer_row=10, rows=50))[0]
    assert d.raster_class(d.Block(dev, row0=0, gny=gny + 1))[0] is False
    # written through the library: the declaration of THAT block is gone, the other one stays
    dev.upload_rows(dem[:100])
    assert d.raster_class(blk)[0] is False and d.raster_class(blk2)[0]
    d.RasterScan().add(blk).add(blk2).declare(blk)
    d.forget_raster_class(blk2)
    assert d.raster_class(blk)[0] and d.raster_class(blk2)[0] is False
    d.dem_changed(dev)
    assert d.raster_class(blk)[0] is False
    d.RasterScan().add(blk).add(blk2).declare(blk, blk2)
    d.forget_raster_class()
    assert d.raster_class(blk)[0] is False and d.raster_class(blk2)[0] is False
    # freed memory takes its declaration along: a new array at the same address starts undeclared
    d.RasterScan().add(blk).add(blk2).declare(blk)
    dev.free()
    again = d.DeviceArray(100, nx)
    assert d.raster_class(d.Block(again, row0=0, gny=gny))[0] is False
    again.free()
    other.free()


def test_a_declaration_made_on_another_thread_applies():
    """The class belongs to the memory: declared by one thread, found by another (a set-up thread and worker threads)."""
    import threading
    size, nb = 19, 3
    _, dem, whole = sensitive_seed(size, nb)
    up, down = shard.halo_rows(_lib.DESC_TPI, size)
    blocks = blocks_of(dem, nb, up, down)
    t = threading.Thread(target=declare, args=(blocks,))
    t.start()
    t.join()
    assert np.array_equal(tpi_of(blocks, size, dem.shape[1]), whole)
    free_blocks(blocks)


@pytest.fixture()
def loopback_comm():
    lib = _lib.lib()
    os.environ["TOPO_AMD_HALO_LOOPBACK"] = "1"
    uid = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
    _lib.check(lib.topo_amd_comm_unique_id(uid), "comm_unique_id")
    _lib.check(lib.topo_amd_comm_init(0, 1, uid.raw), "comm_init")
    try:
        yield lib
    finally:
        _lib.check(lib.topo_amd_comm_destroy(), "comm_destroy")
        _lib.check(lib.topo_amd_shard_layout(-1, -1), "shard_layout")
        os.environ.pop("TOPO_AMD_HALO_LOOPBACK", None)


def test_shard_calls_classify_themselves_through_the_bare_c_abi(loopback_comm):
    """A C caller that has never heard of topo_amd_shard_classify: buffer, upload, topo_amd_shard_tpi_std - and the single
    block's bits on the fractional raster on which undeclared blocks differ (tools/fuzz_repro_r05.py)."""
    lib = loopback_comm
    size = 19
    up, down = shard.halo_rows(_lib.DESC_TPI, size)
    _, local, _ = sensitive_seed(size, 3)
    rows, nx = local.shape
    stacked = np.concatenate([local, local, local], axis=0)
    whole = d.DeviceArray.from_host(stacked)
    want = d.DeviceArray(rows, nx)
    d.Block(whole).tpi_std(size, tpi=want, out_row0=rows, out_rows=rows)
    d.sync()
    want_host = want.to_host()
    # the middle block of the stack as a plain row block with nothing declared: not the whole raster's bits (the test is sharp)
    part = d.DeviceArray.from_host(stacked[rows - up: 2 * rows + down])
    got = d.DeviceArray(rows, nx)
    d.Block(part, row0=rows - up, gny=3 * rows).tpi_std(size, tpi=got, out_row0=rows, out_rows=rows)
    d.sync()
    assert not np.array_equal(got.to_host(), want_host)
    # the shard call, bare C ABI: [up | rows | down] rows, ghost rows poisoned, no layout, no classify
    buf = d.DeviceArray(up + rows + down, nx)
    _lib.check(lib.topo_amd_memset(buf.ptr, 0xFF, buf.nbytes), "memset")
    buf.upload_rows(local, up)
    owned = d.Block(buf, row0=rows, gny=3 * rows, first_buffer_row=up, rows=rows)
    assert d.raster_class(owned)[0] is False
    for _ in range(2):
        _lib.check(lib.topo_amd_shard_tpi_std(buf.ptr, rows, rows, 3 * rows, nx, size, got.ptr, None), "shard_tpi_std")
        d.sync()
        assert np.array_equal(got.to_host(), want_host)
    dec, large, lo, hi, share = d.raster_class(owned)
    assert dec and not large and share > 0.9
    # rewritten through the library: classified again at the next call (another raster in the same buffer)
    other = fractional_window(77)
    other[:, :] = np.rint(other) + 0.5 * (np.arange(other.shape[1]) % 2)  # half-metre steps: another class
    buf.upload_rows(other, up)
    assert d.raster_class(owned)[0] is False
    _lib.check(lib.topo_amd_shard_tpi_std(buf.ptr, rows, rows, 3 * rows, nx, size, got.ptr, None), "shard_tpi_std")
    d.sync()
    whole.upload_rows(np.concatenate([other, other, other], axis=0))
    d.Block(whole).tpi_std(size, tpi=want, out_row0=rows, out_rows=rows)
    d.sync()
    assert np.array_equal(got.to_host(), want.to_host())
    for a in (whole, want, part, got, buf):
        a.free()


def test_shard_classify_refuses_a_partial_shard_without_a_communicator():
    """a ShardedDEM built before init_comm on a multi-rank plan used to declare its own rows' class as the
    raster's."""
    lib = _lib.lib()
    dev = d.DeviceArray.from_host(orc.synthetic_dem(64, 128, seed=1))
    rc = lib.topo_amd_shard_classify(dev.ptr, 64, 64, 192, 128)
    assert rc != 0 and "communicator" in lib.topo_amd_last_error().decode()
    assert lib.topo_amd_shard_classify(dev.ptr, 64, 0, 64, 128) == 0  # the shard is the raster
    dev.free()


def test_scaled_tpi_leaves_samples_its_unit_cannot_hold_to_the_general_kernel():
    """the scaled route stages (int)rint(x * unit); with unit = 2^16 (a raster of small values by its
    class) a sample beyond 2^14 saturates the conversion, and a window made of such samples only has range 0 and used to pass
    the unwrapping test: TPI 7232 m instead of 0 inside a plateau of 40000.  The class can be wrong about a patch (a lattice
    sample, a caller's declaration): here it is declared for a raster of values in [0, 1)."""
    size = 19
    gny, nx = 400, 512
    rng = np.random.default_rng(9)
    dem = rng.random((gny, nx)).astype(np.float32)                 # a normalised surface
    dem[120:260, 100:400] = 40000.0                                # the plateau the class does not know of
    dem[300:330, 50:90] = 20000.0 + rng.random((30, 40)).astype(np.float32)  # beyond 2^14 with fractional parts
    want = orc.tpi_exact(dem, size)
    up, down = shard.halo_rows(_lib.DESC_TPI, size)
    blocks = blocks_of(dem, 2, up, down)
    for blk, _, _ in blocks:
        _lib.check(_lib.lib().topo_amd_raster_class_set(blk.data.row_ptr(blk.first), blk.rows, gny, nx, 0, 0.0, 1.0, 1.0),
                   "raster_class_set")
    got = tpi_of(blocks, size, nx)
    free_blocks(blocks)
    # exact chains (2^-16 m) wherever the scaled one steps aside; 2^-17 m per sample where it runs
    assert float(np.max(np.abs(got - want))) <= 2.5e-4 * 40000.0 / 4096.0 + 2.0 ** -16, float(np.max(np.abs(got - want)))
    assert float(np.max(np.abs(got[130:250, 110:390]))) == 0.0     # inside the plateau
    # and the honest class (the library's own scan of the whole raster) agrees
    assert float(np.max(np.abs(topo.tpi(dem, size) - want))) <= 2.5e-4 * 40000.0 / 4096.0 + 2.0 ** -9
