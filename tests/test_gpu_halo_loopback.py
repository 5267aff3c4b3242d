"""The N > 1 ghost-row exchange, executed on ONE GPU (task 6).

With TOPO_AMD_HALO_LOOPBACK=1 and a communicator of one rank, ``topo_amd_halo_exchange_start`` issues
its real ``ncclSend`` / ``ncclRecv`` pairs to rank 0 itself, with periodic wrap: the block's last
``halo_above`` rows land in its top ghost rows, its first ``halo_below`` rows in its bottom ghost rows.
Pointer offsets, element counts, the ``input_ready`` / ``halo_done`` stream ordering and the 16-CU
reservation of ``run_overlapped`` therefore run on hardware.  A middle shard whose two neighbours are
itself is the middle third of the DEM stacked three times, which gives the sharded entry points an
exact single-block answer to be compared with, bit for bit.  (Precedent for independent blocks plus
ghost rows in the reference: ``map_overlap(..., boundary="none")``, topo.py:177-178.)"""
import os

import numpy as np
import pytest

from oracle import topo_oracle as orc

pytestmark = pytest.mark.gpu

from topo_descriptors_amd import _lib, device as d, shard  # noqa: E402


@pytest.fixture()
def loopback_comm():
    lib = _lib.lib()
    os.environ["TOPO_AMD_HALO_LOOPBACK"] = "1"
    import ctypes as C
    uid = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
    _lib.check(lib.topo_amd_comm_unique_id(uid), "comm_unique_id")
    _lib.check(lib.topo_amd_comm_init(0, 1, uid.raw), "comm_init")
    try:
        yield lib
    finally:
        _lib.check(lib.topo_amd_comm_destroy(), "comm_destroy")
        _lib.check(lib.topo_amd_shard_layout(-1, -1), "shard_layout")
        os.environ.pop("TOPO_AMD_HALO_LOOPBACK", None)


def test_ghosts_are_the_wrapped_rows_three_steps(loopback_comm):
    lib = loopback_comm
    rows, nx, above, below = 600, 2048, 33, 17
    buf = d.DeviceArray(above + rows + below, nx)
    tpi = d.DeviceArray(rows, nx)
    for step in range(3):
        local = orc.synthetic_dem(rows, nx, seed=10 + step)
        # poison the ghost rows, then put this step's rows in place
        _lib.check(lib.topo_amd_memset(buf.ptr, 0xFF, buf.nbytes), "memset")
        buf.upload_rows(local, above)
        _lib.check(lib.topo_amd_halo_exchange_start(buf.ptr, rows, nx, above, below), "halo_exchange_start")
        # an interior launch on the owned rows while the exchange is in flight (what run_overlapped does)
        blk = d.Block(buf, row0=0, gny=rows, first_buffer_row=above, rows=rows)
        blk.tpi_std(67, tpi=tpi)
        _lib.check(lib.topo_amd_halo_wait(), "halo_wait")
        d.sync()
        got = buf.to_host()
        assert np.array_equal(got[above : above + rows], local), step
        assert np.array_equal(got[:above], local[rows - above :]), (step, "top ghost rows = the block's last rows")
        assert np.array_equal(got[above + rows :], local[:below]), (step, "bottom ghost rows = the block's first rows")
    buf.free()
    tpi.free()


@pytest.mark.parametrize("size", [7, 67])
def test_middle_shard_of_a_stacked_dem(loopback_comm, size):
    """shard_tpi_std on a shard whose neighbours are itself == the middle third of the DEM stacked three times."""
    rows, nx = 512, 1024
    local = orc.synthetic_dem(rows, nx, seed=3)
    stacked = np.concatenate([local, local, local], axis=0)
    up, down = shard.halo_rows(_lib.DESC_TPI, size)
    # the plan reserves deeper ghost zones than this descriptor needs: topo_amd_shard_layout
    plan = shard.RowShardPlan(3 * rows, nx, 3, 1, up + 5, down + 9)
    sd = shard.ShardedDEM(plan, local)
    t, s = d.DeviceArray(rows, nx), d.DeviceArray(rows, nx)
    for _ in range(2):
        sd.tpi_std(size, tpi=t, std=s)
    d.sync()
    whole = d.DeviceArray.from_host(stacked)
    wt, ws = d.DeviceArray(rows, nx), d.DeviceArray(rows, nx)
    d.Block(whole).tpi_std(size, tpi=wt, std=ws, out_row0=rows, out_rows=rows)
    d.sync()
    assert np.array_equal(t.to_host(), wt.to_host())
    assert np.array_equal(s.to_host(), ws.to_host())
    for a in (t, s, wt, ws, whole, sd.block):
        a.free()


def test_a_plan_too_shallow_for_the_descriptor_is_refused(loopback_comm):
    rows, nx = 256, 512
    plan = shard.RowShardPlan(3 * rows, nx, 3, 1, 16, 16)  # sized for 33 px
    sd = shard.ShardedDEM(plan, orc.synthetic_dem(rows, nx, seed=1))
    t = d.DeviceArray(rows, nx)
    with pytest.raises(_lib.TopoAmdError, match="ghost rows"):
        sd.tpi_std(67, tpi=t)
    sd.tpi_std(33, tpi=t)
    d.sync()
    t.free()
    sd.block.free()
