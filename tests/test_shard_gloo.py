"""The N > 1 path on CPU: world_size 2 and 3 over gloo.

What runs here is the host logic of the row sharding (``RowShardPlan``: who owns which rows,
which rows travel where, what is interior and what is seam) with torch.distributed/gloo as
the transport in place of RCCL, and the oracle as the per-block evaluator in place of the HIP
kernels.  The claim under test is the one the GPU path relies on: a block plus its ghost
rows, with the boundary rule applied at the global edges only, reproduces the single-domain
result exactly on the owned rows."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from oracle import topo_oracle as orc  # noqa: E402
from topo_descriptors_amd import _lib  # noqa: E402
from topo_descriptors_amd.shard import RowShardPlan, halo_rows, split_rows, sx_multi_halo  # noqa: E402


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def exchange(block, plan):
    """The protocol of topo_amd_halo_exchange_start with gloo point-to-point calls."""
    first = plan.halo_above
    reqs = []
    keep = []
    for peer, row, rows in plan.sends():
        t = torch.from_numpy(np.ascontiguousarray(block[first + row: first + row + rows]))
        keep.append(t)
        reqs.append(dist.isend(t, dst=peer))
    bufs = []
    for peer, row, rows in plan.recvs():
        t = torch.empty((rows, plan.nx), dtype=torch.float32)
        bufs.append((row, rows, t))
        reqs.append(dist.irecv(t, src=peer))
    for r in reqs:
        r.wait()
    for row, rows, t in bufs:
        block[row: row + rows] = t.numpy()


def evaluate(name, window, x, y_window, params):
    """Oracle on a haloed block.  Block edges that are not global edges only corrupt rows
    inside the ghost depth, which are cropped away by the caller."""
    if name == "tpi":
        return orc.tpi_exact(window, params["size"])
    if name == "std":
        return orc.std_exact(window, params["size"])
    if name == "gauss":
        return orc.gaussian_exact(window, params["sigma"])
    if name == "slope":
        res = orc.grid_resolution(x, y_window)
        return orc.gradient_exact(window, params["sigma"], res)[2]
    if name == "sx":
        # the oracle zeroes a frame of `w` rows at BOTH ends of whatever it is given; at an open
        # seam that frame must fall on throw-away rows, so pad `w` rows there and crop again
        # several azimuths (topo_amd_shard_sx_multi): one block with the union of the ghost rows
        # serves every sector; the planes are stacked along x here
        planes = []
        for az in params.get("azimuths", [0.0]):
            w, _, _ = orc.sx_geometry(az, params["radius"], 30.0, -30.0)
            top = 0 if params["at_top"] else w
            bot = 0 if params["at_bottom"] else w
            padded = np.pad(window, ((top, bot), (0, 0)))
            yy = np.concatenate([y_window[0] + 30.0 * np.arange(top, 0, -1), y_window,
                                 y_window[-1] - 30.0 * np.arange(1, bot + 1)])
            out = orc.sx(padded, x, yy, az, params["radius"]).astype(np.float64)
            planes.append(out[top: top + window.shape[0]])
        return np.concatenate(planes, axis=1)
    if name == "valley":
        # standardised with the statistics of the WHOLE DEM, which the ranks obtained by an all-reduce
        return orc.valley_ridge_exact(window, params["size"], "valley", angles=params["angles"],
                                      stats=params["stats"], method="direct")[0]
    raise KeyError(name)


CASES = [("tpi", {"size": 17}, _lib.DESC_TPI), ("tpi", {"size": 6}, _lib.DESC_TPI),
         ("std", {"size": 7}, _lib.DESC_STD), ("gauss", {"sigma": 2.25}, _lib.DESC_GAUSS),
         ("slope", {"sigma": 2.25}, _lib.DESC_GRADIENT), ("slope", {"sigma": 0.75}, _lib.DESC_GRADIENT),
         ("sx", {"radius": 150.0}, _lib.DESC_SX),
         ("sx", {"radius": 150.0, "azimuths": [0.0, 5.0, 180.0, 270.0]}, _lib.DESC_SX),
         ("valley", {"size": 7, "angles": np.array([0, 30, 45, 100, 160], dtype=np.float32)},
          _lib.DESC_VALLEY_RIDGE)]


def worker(rank, world, port, gny, nx, fail):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dem = orc.synthetic_dem(gny, nx, seed=21)
        x = 2600000.0 + 30.0 * np.arange(nx)
        y = 1200000.0 - 30.0 * np.arange(gny)
        for name, params, desc in CASES:
            if desc == _lib.DESC_SX:
                sectors = []
                for az in params.get("azimuths", [0.0]):
                    window, offs, dist_m = orc.sx_geometry(az, params["radius"], 30.0, -30.0)
                    sectors.append((window, offs[:, 0], offs[:, 1], dist_m))
                up, down = sx_multi_halo(sectors)   # the union of the sectors' reach
                assert (up, down) == halo_rows(desc, max(max(0, -s[1].min()) for s in sectors),
                                               max(max(0, s[1].max()) for s in sectors))
                if "azimuths" in params:
                    assert up > 0 and down > 0      # north and south rays in one exchange
            elif desc == _lib.DESC_VALLEY_RIDGE:
                kmax = max(orc.rotate_kernels(orc.valley_kernels(params["size"], [0, 0.15, 0.3]), a).shape[1]
                           for a in params["angles"])
                up, down = halo_rows(desc, kmax)
                assert (up, down) == (kmax // 2, kmax - 1 - kmax // 2)
            else:
                p0 = params.get("size", params.get("sigma"))
                up, down = halo_rows(desc, p0)
            plan = RowShardPlan(gny, nx, world, rank, up, down)
            plan.validate()
            block = np.full((plan.buffer_rows, nx), np.nan, dtype=np.float32)
            block[up: up + plan.rows_local] = dem[plan.row0: plan.row0 + plan.rows_local]
            exchange(block, plan)
            first, rows, g0 = plan.valid_window()
            window = block[first: first + rows]
            assert not np.isnan(window).any()
            assert np.array_equal(window, dem[g0: g0 + rows])        # ghosts are the neighbours' rows
            if desc == _lib.DESC_VALLEY_RIDGE:
                # the protocol of topo_amd_shard_valley_ridge: float64 moments about 0 of the owned
                # rows, one all-reduce, mean and population std of the whole DEM on every rank
                own = dem[plan.row0: plan.row0 + plan.rows_local].astype(np.float64)
                mom = torch.tensor([own.size, own.sum(), (own * own).sum()], dtype=torch.float64)
                dist.all_reduce(mom)
                cnt, s1, s2 = (float(v) for v in mom)
                mean = s1 / cnt
                stats = (mean, float(np.sqrt(max(s2 / cnt - mean * mean, 0.0))))
                full = dem.astype(np.float64)
                assert cnt == dem.size and s1 == full.sum() and s2 == (full * full).sum()  # exact on whole metres
                params = dict(params, stats=stats)
            params = dict(params, at_top=(g0 == 0), at_bottom=(g0 + rows == gny))
            out = evaluate(name, window, x, y[g0: g0 + rows], params)
            mine = out[plan.row0 - g0: plan.row0 - g0 + plan.rows_local]
            whole = evaluate(name, dem, x, y, dict(params, at_top=True, at_bottom=True))
            want = whole[plan.row0: plan.row0 + plan.rows_local]
            assert np.array_equal(mine, want), (name, params, rank)
            a, b = plan.interior()
            assert plan.row0 <= a <= b <= plan.row0 + plan.rows_local
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:  # noqa: BLE001
        fail.put(f"rank {rank}: {type(exc).__name__}: {exc}")
        raise


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharding_reproduces_single_domain(world):
    ctx = mp.get_context("spawn")
    fail = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, 96, 80, fail)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    errors = []
    while not fail.empty():
        errors.append(fail.get())
    assert not errors, errors
    assert all(p.exitcode == 0 for p in procs)


def test_plan_geometry():
    assert split_rows(10, 3) == [(0, 4), (4, 3), (7, 3)]
    p = RowShardPlan(32768, 32768, 8, 3, 33, 33)
    assert (p.row0, p.rows_local, p.buffer_rows) == (12288, 4096, 4162)
    assert p.sends() == [(2, 0, 33), (4, 4096 - 33, 33)]
    assert p.recvs() == [(2, 0, 33), (4, 33 + 4096, 33)]
    assert p.interior() == (12288 + 33, 12288 + 4096 - 33)
    top = RowShardPlan(32768, 32768, 8, 0, 33, 33)
    assert top.sends() == [(1, 4096 - 33, 33)] and top.recvs() == [(1, 33 + 4096, 33)]
    assert top.valid_window() == (33, 4096 + 33, 0)
    sx = RowShardPlan(1000, 64, 4, 2, 17, 0)           # one-sided ghost zone (Sx, azimuth 0)
    assert sx.sends() == [(3, 250 - 17, 17)] and sx.recvs() == [(1, 0, 17)]
    with pytest.raises(ValueError):
        RowShardPlan(100, 8, 8, 0, 33, 33).validate()   # 12-row shards cannot feed 33 ghost rows


def _rendezvous_worker(rank, world, port, fail):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                          WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        import bench
        rdv = bench.Rendezvous(world)
        assert rdv.rank == rank
        payload = rdv.bcast_bytes(bytes(range(128)) if rank == 0 else None)
        assert payload == bytes(range(128))       # the ncclUniqueId travels like this
        rdv.barrier()
        assert rdv.max(float(rank)) == float(world - 1)
        # the per-descriptor table of the sharded run (BASELINE configs[4]): a collective over the same control
        # plane; the slowest rank's median counts.  Stand-ins for the GPU steps and their HIP-event timer.
        calls = []
        steps = {k: (lambda k=k: calls.append(k)) for k in bench.SHARD_KEYS}

        def fake_timer(fn, reps, warm):
            for _ in range(reps + warm):
                fn()
            return [1.0 + rank + 0.01 * i for i in range(reps)]

        table = bench.sharded_descriptors(rdv, steps, fake_timer, 32768 * 32768, world, reps=4)
        assert tuple(table) == bench.SHARD_KEYS and calls == [k for k in bench.SHARD_KEYS for _ in range(6)]
        for key, row in table.items():
            assert abs(row["ms"] - (world + 0.015)) < 1e-9, row        # median of the slowest rank
            assert row["ms_min"] == 1.0 and abs(row["ms_max"] - (world + 0.03)) < 1e-9
            assert row["Mpixels_per_s"] == round(32768 * 32768 / row["ms"] / 1e3, 1)
            assert row["entry_point"].startswith("topo_amd_shard_")
            assert 0 < row["hbm_frac"] < 1
        rdv.close()
    except Exception as exc:  # noqa: BLE001
        fail.put(f"rank {rank}: {type(exc).__name__}: {exc}")
        raise


def test_bench_rendezvous_over_gloo():
    """The control plane bench.py uses for N > 1 (id broadcast, barrier, max over ranks)."""
    ctx = mp.get_context("spawn")
    fail = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_rendezvous_worker, args=(r, 2, port, fail)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    errors = []
    while not fail.empty():
        errors.append(fail.get())
    assert not errors, errors
    assert all(p.exitcode == 0 for p in procs)
