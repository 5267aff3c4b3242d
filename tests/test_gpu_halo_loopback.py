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


# ---- every topo_amd_shard_* entry point under the live exchange (task 1) ---------------------------
def _middle_shard(local, up, down, extra=(0, 0)):
    """ShardedDEM of the middle third of `local` stacked three times, ghost rows poisoned (0xFF = NaN bit pattern)
    so that a row the exchange fails to deliver shows up in the outputs."""
    rows, nx = local.shape
    plan = shard.RowShardPlan(3 * rows, nx, 3, 1, up + extra[0], down + extra[1])
    sd = shard.ShardedDEM(plan)
    _lib.check(_lib.lib().topo_amd_memset(sd.block.ptr, 0xFF, sd.block.nbytes), "memset")
    sd.block.upload_rows(local, plan.halo_above)
    return sd


def _poison_ghosts(sd):
    p = sd.plan
    lib = _lib.lib()
    if p.halo_above:
        _lib.check(lib.topo_amd_memset(sd.block.ptr, 0xFF, p.halo_above * p.nx * 4), "memset")
    if p.halo_below:
        _lib.check(lib.topo_amd_memset(sd.block.row_ptr(p.halo_above + p.rows_local), 0xFF, p.halo_below * p.nx * 4),
                   "memset")


# sigma 0.75: Sobel, 1 ghost row.  2.25 / 3.25: matrix-core route below radius 16, 17 ghost rows (ADVICE r02, high:
# the shard used to be laid out with R + 1).  1.5: vector-ALU route (radius 6).  7.0: matrix cores, radius 28.
# (3.25, 2.0): anisotropic, two smooths.  30.25 on 8800 local rows: row chunks + the epilogue on the second stream
# inside run_overlapped.
@pytest.mark.parametrize("sigma,ratio,rows,nx", [(0.75, 1.0, 256, 512), (1.5, 1.0, 256, 512), (2.25, 1.0, 300, 512),
                                                 (3.25, 1.0, 300, 512), (3.25, 1.0, 263, 516), (7.0, 1.0, 300, 512),
                                                 (3.25, 2.0, 300, 512), (30.25, 1.0, 8800, 512)])
def test_shard_gradient_under_the_live_exchange(loopback_comm, sigma, ratio, rows, nx):
    local = orc.synthetic_dem(rows, nx, seed=17)
    stacked = np.concatenate([local, local, local], axis=0)
    up, down = shard.halo_rows(_lib.DESC_GRADIENT, sigma, ratio)
    x = 2600000.0 + 30.0 * np.arange(nx)
    y = 1200000.0 - 30.0 * np.arange(3 * rows)
    res = orc.grid_resolution(x, y)
    sd = _middle_shard(local, up, down, extra=(3, 0) if rows < 1000 else (0, 0))
    outs = [d.DeviceArray(rows, nx) for _ in range(4)]
    for _ in range(2):
        _poison_ghosts(sd)
        sd.gradient(sigma, res["x"], res["y"], sig_ratio=ratio, dx=outs[0], dy=outs[1], slope=outs[2], aspect=outs[3])
    d.sync()
    whole = d.DeviceArray.from_host(stacked)
    want = [d.DeviceArray(rows, nx) for _ in range(4)]
    d.Block(whole).gradient(sigma, res["x"], res["y"], sig_ratio=ratio, dx=want[0], dy=want[1], slope=want[2],
                            aspect=want[3], out_row0=rows, out_rows=rows)
    d.sync()
    for k, name in enumerate(("dx", "dy", "slope", "aspect")):
        assert np.array_equal(outs[k].to_host(), want[k].to_host()), (sigma, ratio, name)
    for a in outs + want + [whole, sd.block]:
        a.free()


def test_c_caller_with_the_default_layout_gets_the_same_gradient(loopback_comm):
    """A C caller that lays its block out with exactly topo_amd_halo_rows(GRADIENT) ghost rows and never calls
    topo_amd_shard_layout."""
    lib = loopback_comm
    rows, nx, sigma = 288, 512, 3.25
    local = orc.synthetic_dem(rows, nx, seed=23)
    stacked = np.concatenate([local, local, local], axis=0)
    up, down = shard.halo_rows(_lib.DESC_GRADIENT, sigma, 1.0)
    assert (up, down) == (17, 17)
    buf = d.DeviceArray(up + rows + down, nx)
    _lib.check(lib.topo_amd_memset(buf.ptr, 0xFF, buf.nbytes), "memset")
    buf.upload_rows(local, up)
    rx, ry = np.array([30.0]), np.array([-30.0])
    outs = [d.DeviceArray(rows, nx) for _ in range(4)]
    _lib.check(lib.topo_amd_shard_gradient(buf.ptr, rows, rows, 3 * rows, nx, sigma, 1.0, _lib.RES_SCALAR, _lib.ptr(rx),
                                           _lib.ptr(ry), *[o.ptr for o in outs]), "shard_gradient")
    d.sync()
    whole = d.DeviceArray.from_host(stacked)
    want = [d.DeviceArray(rows, nx) for _ in range(4)]
    d.Block(whole).gradient(sigma, rx, ry, dx=want[0], dy=want[1], slope=want[2], aspect=want[3], out_row0=rows,
                            out_rows=rows)
    d.sync()
    for k in range(4):
        assert np.array_equal(outs[k].to_host(), want[k].to_host()), k
    for a in outs + want + [whole, buf]:
        a.free()


@pytest.mark.parametrize("azimuth", [0.0, 180.0, 135.0, 275.0])
def test_shard_sx_under_the_live_exchange(loopback_comm, azimuth):
    """One-sided ghost zones: at azimuth 0 on a north-up grid only the north neighbour sends."""
    rows, nx = 320, 512
    local = orc.synthetic_dem(rows, nx, seed=29)
    stacked = np.concatenate([local, local, local], axis=0)
    window, dj, di, dist = d.sx_offsets(azimuth, 500.0, 30.0, -30.0)
    up, down = shard.halo_rows(_lib.DESC_SX, max(0, -dj.min()), max(0, dj.max()))
    if azimuth == 0.0:
        assert up > 0 and down == 0
    if azimuth == 180.0:
        assert up == 0 and down > 0
    sd = _middle_shard(local, up, down)
    out = d.DeviceArray(rows, nx)
    for _ in range(2):
        _poison_ghosts(sd)
        sd.sx(dj, di, dist, window, 10.0, out)
    d.sync()
    whole = d.DeviceArray.from_host(stacked)
    want = d.DeviceArray(rows, nx)
    d.Block(whole).sx(dj, di, dist, window, 10.0, want, out_row0=rows, out_rows=rows)
    d.sync()
    got, ref = out.to_host(), want.to_host()
    assert np.array_equal(got, ref), azimuth
    assert np.isfinite(got).all()
    for a in (out, want, whole, sd.block):
        a.free()


def test_shard_sx_multi_north_and_south_rays_in_one_exchange(loopback_comm):
    rows, nx = 320, 512
    local = orc.synthetic_dem(rows, nx, seed=31)
    stacked = np.concatenate([local, local, local], axis=0)
    sectors = [d.sx_offsets(a, 400.0, 30.0, -30.0) for a in (350.0, 0.0, 10.0, 170.0, 180.0, 190.0, 90.0)]
    up, down = shard.sx_multi_halo(sectors)
    assert up > 0 and down > 0
    sd = _middle_shard(local, up, down)
    outs = [d.DeviceArray(rows, nx) for _ in sectors]
    for _ in range(2):
        _poison_ghosts(sd)
        sd.sx_multi(sectors, 10.0, outs)
    d.sync()
    whole = d.DeviceArray.from_host(stacked)
    blk = d.Block(whole)
    for sec, o in zip(sectors, outs):
        want = d.DeviceArray(rows, nx)
        blk.sx(sec[1], sec[2], sec[3], sec[0], 10.0, want, out_row0=rows, out_rows=rows)
        d.sync()
        assert np.array_equal(o.to_host(), want.to_host())
        want.free()
        o.free()
    whole.free()
    sd.block.free()


@pytest.mark.parametrize("route", ["direct", "fft"])
def test_shard_valley_ridge_under_the_live_exchange(loopback_comm, route, monkeypatch):
    """The moments of the middle third are those of the stack (the same samples three times; whole metres: exact),
    so the standardisation agrees and the direct kernel must give the single block's bits."""
    from topo_descriptors_amd import topo
    monkeypatch.setenv("TOPO_AMD_VALLEY_FFT_MIN_KERNEL", "1" if route == "fft" else "100000")
    rows, nx = 160, 256
    local = orc.synthetic_dem(rows, nx, seed=37)
    stacked = np.concatenate([local, local, local], axis=0)
    size, flats = 7, [0, 0.15, 0.3]
    angles_in = np.arange(0, 180, 9, dtype=np.float32)
    taps, ksize, angles = topo._valley_ridge_tables(topo._valley_kernels(size, flats), angles_in)
    up, down = shard.halo_rows(_lib.DESC_VALLEY_RIDGE, int(ksize.max()))
    sd = _middle_shard(local, up, down)
    n, a = d.DeviceArray(rows, nx), d.DeviceArray(rows, nx)
    for _ in range(2):
        _poison_ghosts(sd)
        sd.valley_ridge(taps, ksize, angles, len(flats), n, a)
    d.sync()
    whole = d.DeviceArray.from_host(stacked)
    mean, stdev = d.mean_std(whole)
    n2, a2 = d.DeviceArray(rows, nx), d.DeviceArray(rows, nx)
    d.Block(whole).valley_ridge(taps, ksize, angles, len(flats), mean, stdev, n2, a2, out_row0=rows, out_rows=rows)
    d.sync()
    norm, direction = n.to_host(), a.to_host()
    assert np.isfinite(norm).all()
    if route == "direct":
        assert np.array_equal(norm, n2.to_host()) and np.array_equal(direction, a2.to_host())
    else:
        scale = float(np.max(np.abs(n2.to_host())))
        assert np.max(np.abs(norm - n2.to_host())) <= 1e-5 * scale
        assert np.mean(direction == a2.to_host()) >= 0.995
    for x in (n, a, n2, a2, whole, sd.block):
        x.free()


def test_shard_tpi_std_fractional_dem_under_the_live_exchange(loopback_comm):
    """Fractional elevations take other kernels (fraction march / general kernel) than whole metres: the seam
    strips must agree with the single block there too."""
    rows, nx, size = 384, 768, 33
    local = orc.synthetic_dem(rows, nx, seed=41, integer=False)
    stacked = np.concatenate([local, local, local], axis=0)
    up, down = shard.halo_rows(_lib.DESC_TPI, size)
    sd = _middle_shard(local, up, down)
    t, s = d.DeviceArray(rows, nx), d.DeviceArray(rows, nx)
    for _ in range(2):
        _poison_ghosts(sd)
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


# ---- round 4: one launch per kernel behind the ghost-row gate, in every mode (task 1) ---------------
_GATE_CHILD = r"""
import os, sys, zlib, json
import numpy as np
sys.path.insert(0, %r)
os.environ["TOPO_AMD_HALO_LOOPBACK"] = "1"
from oracle import topo_oracle as orc
from topo_descriptors_amd import _lib, device as d, shard
import ctypes as C
lib = _lib.lib()
shard.ShardedDEM.init_comm(0, 1, lambda payload: payload)
rows, nx = 2112, 16384           # 90 strips x 36 tile rows of the 67-px kernels: more tiles than blocks, every block has a run
local = orc.synthetic_dem(rows, nx, seed=21)
local[700:760, 3000:3300] += 0.25   # a patch of fractional elevations: the second kernels of the chains have work
deep = max(shard.halo_rows(_lib.DESC_GRADIENT, 30.25, 1.0))
plan = shard.RowShardPlan(3 * rows, nx, 3, 1, deep, deep)
sd = shard.ShardedDEM(plan)
outs = [d.DeviceArray(rows, nx) for _ in range(4)]
window, dj, di, dist = d.sx_offsets(0.0, 500.0, 30.0, -30.0)
w2, dj2, di2, dist2 = d.sx_offsets(135.0, 300.0, 30.0, -30.0)
crc = {}
def poison():
    p = sd.plan
    _lib.check(lib.topo_amd_memset(sd.block.ptr, 0xFF, p.halo_above * p.nx * 4), "memset")
    _lib.check(lib.topo_amd_memset(sd.block.row_ptr(p.halo_above + p.rows_local), 0xFF, p.halo_below * p.nx * 4), "memset")
def take(name, planes):
    d.sync()
    crc[name] = [zlib.crc32(outs[k].to_host().tobytes()) for k in range(planes)]
sd.block.upload_rows(local, plan.halo_above)
for rep in range(5):             # (auto mode: careful calls first, lean ones after three clean careful calls)
    poison(); sd.tpi_std(67, tpi=outs[0], std=outs[1]); take("tpi_std67_%%d" %% rep, 2)
poison(); sd.tpi_std(67, tpi=outs[0]); take("tpi67", 1)
poison(); sd.tpi_std(7, tpi=outs[0], std=outs[1]); take("tpi_std7", 2)
poison(); sd.gradient(3.25, [30.0], [-30.0], dx=outs[0], dy=outs[1], slope=outs[2], aspect=outs[3]); take("grad3", 4)
poison(); sd.gradient(30.25, [30.0], [-30.0], dx=outs[0], dy=outs[1], slope=outs[2], aspect=outs[3]); take("grad30", 4)
poison(); sd.sx(dj, di, dist, window, 10.0, outs[0]); take("sx0", 1)
poison(); sd.sx(dj2, di2, dist2, w2, 10.0, outs[0]); take("sx135", 1)
gave_up = C.c_uint()
_lib.check(lib.topo_amd_gate_giveups(C.byref(gave_up)), "gate_giveups")
print(json.dumps({"crc": crc, "gave_up": gave_up.value}))
"""


def test_gate_modes_give_the_single_block_bits(tmp_path):
    """Sharded TPI / STD, gradient and Sx through (a) the three launches of rounds 1-3, (b) one launch behind the gate
    with the clean-up launch (careful), (c) without it (lean), (d) the default (careful, then lean): every plane has
    the same CRC-32 in all four, and those of the single block."""
    import json
    import subprocess
    import sys
    import zlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    results = {}
    for name, env in (("three_launches", {"TOPO_AMD_SHARD_FUSED": "0"}), ("careful", {"TOPO_AMD_GATE_MODE": "careful"}),
                      ("lean", {"TOPO_AMD_GATE_MODE": "lean"}), ("auto", {"TOPO_AMD_GATE_MODE": "auto"}), ("default", {})):
        out = subprocess.run([sys.executable, "-c", _GATE_CHILD % root], cwd=root, env=dict(os.environ, **env),
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (name, out.stderr[-3000:])
        results[name] = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    base = results["three_launches"]["crc"]
    for name in ("careful", "lean", "auto", "default"):
        assert results[name]["crc"] == base, name
    assert results["lean"]["gave_up"] == 0  # (lean blocks wait; a wait that runs out is an error, not a statistic)
    # and against the single block, in this process
    rows, nx = 2112, 16384
    local = orc.synthetic_dem(rows, nx, seed=21)
    local[700:760, 3000:3300] += 0.25
    whole = d.DeviceArray.from_host(np.concatenate([local, local, local], axis=0))
    outs = [d.DeviceArray(rows, nx) for _ in range(4)]
    blk = d.Block(whole)
    blk.tpi_std(67, tpi=outs[0], std=outs[1], out_row0=rows, out_rows=rows)
    d.sync()
    assert [zlib.crc32(outs[k].to_host().tobytes()) for k in range(2)] == base["tpi_std67_0"]
    blk.gradient(3.25, [30.0], [-30.0], dx=outs[0], dy=outs[1], slope=outs[2], aspect=outs[3], out_row0=rows, out_rows=rows)
    d.sync()
    assert [zlib.crc32(outs[k].to_host().tobytes()) for k in range(4)] == base["grad3"]
    window, dj, di, dist = d.sx_offsets(0.0, 500.0, 30.0, -30.0)
    blk.sx(dj, di, dist, window, 10.0, outs[0], out_row0=rows, out_rows=rows)
    d.sync()
    assert zlib.crc32(outs[0].to_host().tobytes()) == base["sx0"][0]
    for a in outs + [whole]:
        a.free()
