"""Row-shard efficiency on ONE GPU: the step of one 4096-row shard of the 8-way split of the 32768^2 DEM, with its
ghost rows refreshed through the live RCCL exchange in loop-back (TOPO_AMD_HALO_LOOPBACK=1: the shard is its own
upper and lower neighbour), against one eighth of the single-GPU time of the whole DEM.

    python tools/shard_efficiency.py [keys ...]        (TOPO_AMD_SHARD_FUSED=0: the three-launch route of rounds 1-3)

Prints one JSON object: per key the whole-DEM ms, the shard ms and shard_efficiency = full_ms / (8 x shard_ms).
Both ends of the link are one device, so this is the per-shard step of an 8-GPU run minus the wire."""
import json
import os
import sys

os.environ.setdefault("TOPO_AMD_HALO_LOOPBACK", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from topo_descriptors_amd import _lib, device as d, shard as shard_mod  # noqa: E402

NY = NX = int(os.environ.get("SHARD_EFF_N", "32768"))
PARTS = 8
REPS = int(os.environ.get("SHARD_EFF_REPS", "12"))
KEYS = ("tpi_s67", "std_s67", "tpi_std_s67", "std_s7", "tpi_std_s7", "gradient_sigma3.25", "gradient_sigma30.25", "sx_az0_r500")


def median(v):
    v = sorted(v)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def steps(target, outs):
    window, dj, di, dist = d.sx_offsets(0.0, 500.0, 30.0, -30.0)
    return {
        "tpi_s67": lambda: target.tpi_std(67, tpi=outs[0]),
        "std_s67": lambda: target.tpi_std(67, std=outs[1]),
        "tpi_std_s67": lambda: target.tpi_std(67, tpi=outs[0], std=outs[1]),
        "std_s7": lambda: target.tpi_std(7, std=outs[1]),
        "tpi_std_s7": lambda: target.tpi_std(7, tpi=outs[0], std=outs[1]),
        "gradient_sigma3.25": lambda: target.gradient(3.25, [30.0], [-30.0], dx=outs[0], dy=outs[1], slope=outs[2], aspect=outs[3]),
        "gradient_sigma30.25": lambda: target.gradient(30.25, [30.0], [-30.0], dx=outs[0], dy=outs[1], slope=outs[2], aspect=outs[3]),
        "sx_az0_r500": lambda: target.sx(dj, di, dist, window, 10.0, outs[0]),
    }


def main():
    keys = [k for k in sys.argv[1:] if k in KEYS] or list(KEYS)
    _lib.lib()
    shard_mod.ShardedDEM.init_comm(0, 1, lambda payload: payload)
    rows = NY // PARTS
    # the shard first (its buffers are small), then the whole DEM
    deep = max(shard_mod.halo_rows(_lib.DESC_GRADIENT, 30.25, 1.0))
    plan = shard_mod.RowShardPlan(3 * rows, NX, 3, 1, deep, deep)  # the middle shard of three; its neighbours are itself
    sd = shard_mod.ShardedDEM(plan)
    d.synth_dem(rows, NX, row0=plan.row0, seed=0, out=sd.block, out_row=plan.halo_above)
    outs = [d.DeviceArray(rows, NX) for _ in range(4)]
    d.sync()
    shard_ms = {}
    fns = steps(sd, outs)
    for rnd in range(2):  # two rounds: the second one with the clocks up
        for k in keys:
            shard_ms[k] = median(d.time_launches(fns[k], REPS, 3))
    d.sync()
    import ctypes
    gave_up = ctypes.c_uint()
    _lib.check(_lib.lib().topo_amd_gate_giveups(ctypes.byref(gave_up)), "gate_giveups")
    for o in outs:
        o.free()
    sd.block.free()
    full = d.synth_dem(NY, NX, seed=0)
    outs = [d.DeviceArray(NY, NX) for _ in range(4)]
    blk = d.Block(full)
    fns = steps(blk, outs)
    full_ms = {}
    for rnd in range(2):
        for k in keys:
            full_ms[k] = median(d.time_launches(fns[k], max(4, REPS // 2), 2))
    d.sync()
    # the same 4096 output rows as a plain row block of the whole DEM (ghost rows read from the DEM itself, no
    # exchange, no seams, full grid): what a launch of this size costs before any shard mechanics
    class Rows:
        def __init__(self, blk, o0, on):
            self.blk, self.o0, self.on = blk, o0, on

        def tpi_std(self, size, tpi=None, std=None):
            self.blk.tpi_std(size, tpi=tpi, std=std, out_row0=self.o0, out_rows=self.on)

        def gradient(self, sigma, rx, ry, **kw):
            self.blk.gradient(sigma, rx, ry, out_row0=self.o0, out_rows=self.on, **kw)

        def sx(self, dj, di, dist, window, height, out):
            self.blk.sx(dj, di, dist, window, height, out, out_row0=self.o0, out_rows=self.on)

    fns = steps(Rows(blk, rows, rows), outs)
    block_ms = {}
    for rnd in range(2):
        for k in keys:
            block_ms[k] = median(d.time_launches(fns[k], REPS, 3))
    d.sync()
    out = {}
    for k in keys:
        out[k] = {"full_ms": round(full_ms[k], 4), "row_block_ms": round(block_ms[k], 4), "shard_ms": round(shard_ms[k], 4),
                  "shard_efficiency": round(full_ms[k] / (PARTS * shard_ms[k]), 4),
                  "ratio_to_eighth": round(PARTS * shard_ms[k] / full_ms[k], 4)}
    print(json.dumps({"dem": [NY, NX], "shard_rows": rows, "fused": os.environ.get("TOPO_AMD_SHARD_FUSED", "1"),
                      "reserve_cus": os.environ.get("TOPO_AMD_RESERVE_CUS", "default"),
                      "gate_giveups": gave_up.value, "keys": out}))


if __name__ == "__main__":
    main()
