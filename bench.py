#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of the descriptor hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json north_star): TPI at scale 2000 m (disc of 67 px on a 30 m grid) over a
32768 x 32768 synthetic float32 DEM.  One "step" = one pass of the TPI kernel over the whole
DEM.  With N > 1 (launched by torch.distributed.run, one rank per GPU) the DEM is split into
N contiguous row blocks - the total work is fixed (strong scaling) - and every step first
refreshes the 33 ghost rows from the neighbours over RCCL, overlapped with the interior rows.
Inputs are generated on the device and are resident in HBM before the timed region.

With N > 1 the line also carries `descriptors`: STD and TPI+STD at 67 px, the gradient at sigma 3.25 and 30.25 and Sx
(azimuth 0, radius 500 m) through the sharded entry points (topo_amd_shard_*), each with its own ghost-row exchange
per step - BASELINE.json's "per descriptor" metric for configs[4].  TOPO_AMD_HALO_LOOPBACK=1 with one rank
(`python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`) runs the same sharded steps on ONE GPU
as the middle shard of three whose neighbours are itself: the exchange is real (ncclSend / ncclRecv on the
communication stream), the scaling is not.

Rank 0 prints ONE JSON line.  `roofline` prices the TPI kernel at SURVEY.md 8(d)'s 8 B/pixel
(4 read + 4 written) against 8 TB/s (and, as `frac_read_only_basis`, at north_star's literal
"HBM-read" 4 B/pixel); `cpu_baseline` times the oracle's scipy restatement of the reference path
on one host core for a bounded sample.  torch is used for rendezvous, barriers and the
max-over-ranks only; it never touches the GPU.

Timing: `value` and `ms_per_step` come from the wall clock around exactly K steps between two
barriers, as the driver's contract asks.  Every step is also bracketed by its own HIP events
(topo_amd_mark): `ms_per_step_median` / `_mean` / `_min` / `_max` are those per-launch durations,
and the roofline uses their mean.  Before the W warm-up steps the clocks are brought up by untimed
launches until three consecutive ones agree within 1.5 % (`clock_ramp_steps`; the first ~8 launches
after an idle period run up to 25 % slow), so that W = 5 and W = 10 give the same number.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s HBM3E
BYTES_PER_PIXEL = {"tpi": 8, "std": 8, "tpi_std": 12, "gradient": 20, "sx": 8, "gaussian": 8}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    # the GPU clocks are still ramping up during the first ~8 launches after an idle period (5.6 -> 4.45 ms
    # per launch in the kernel trace), hence more untimed steps by default than a cold cache needs
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--ny", type=int, default=32768)
    ap.add_argument("--nx", type=int, default=32768)
    ap.add_argument("--size", type=int, default=67, help="disc diameter in pixels (2000 m / 30 m)")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-descriptor side table")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the host-buffer end_to_end section (it launches the headline kernel on a 16384^2 DEM: under "
                         "rocprofv3 --stats those launches would enter the kernel's average duration)")
    return ap.parse_args()


class Rendezvous:
    """Barrier / broadcast / max over ranks.  Single process: no torch at all."""

    def __init__(self, world):
        self.world = world
        self.rank = int(os.environ.get("RANK", "0"))
        self.dist = None
        if world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=self.rank, world_size=world)
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def bcast_bytes(self, payload):
        if not self.dist:
            return payload
        box = [payload]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def max(self, value):
        if not self.dist:
            return value
        import torch

        t = torch.tensor([value], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


KERNEL_SOURCES = ["disc_wave_impl.hpp", "disc_ring_impl.hpp", "disc_runs.hpp", "disc_wave.hip", "common.hpp"]


def kernel_sources_sha256():
    """Hash of the sources the headline kernel is compiled from: a PMC profile taken on other sources says
    nothing about this build."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(REPO, "topo_descriptors_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def measured_traffic(ny, nx, size, world):
    """HBM bytes per launch of the TPI kernel from the committed rocprofv3 PMC passes
    (profiles/r06_tpi67_traffic.json, made by tools/pmc_passes.sh + tools/traffic_from_pmc.py on this exact
    workload), with the git head and the kernel-source hash the profile was taken at.  The number is nulled
    when the workload or the sources differ from the profiled ones."""
    info = {"traffic": None, "traffic_profile_head": None, "traffic_profile": "profiles/r06_tpi67_traffic.json"}
    path = os.path.join(REPO, "profiles", "r06_tpi67_traffic.json")
    try:
        with open(path) as fh:
            prof = json.load(fh)
    except (OSError, ValueError):
        info["traffic_note"] = "no committed PMC profile"
        return info
    info["traffic_profile_head"] = prof.get("git_head")
    if (ny, nx, size, world) != (32768, 32768, 67, 1):
        info["traffic_note"] = "profiled workload is 32768x32768, 67 px, 1 GPU"
    elif prof.get("kernel_sources_sha256") != kernel_sources_sha256():
        info["traffic_note"] = "kernel sources changed since the PMC passes were taken: re-run tools/pmc_passes.sh"
    else:
        info["traffic"] = prof.get("traffic_bytes_per_launch")
    return info


def valu_bound(ny, nx, size, world, kernel_ms):
    """What bounds the kernel in practice: vector-ALU issue.  profiles/r06_tpi67_valu_bound.json (tools/valu_bound.py)
    prices the kernel's own instruction stream - the row loop's instructions by issue class from the ISA, the rest
    from the launch's SQ_INSTS_VALU counter - with the issue costs measured on the GPU; the fraction is that time
    over the measured one.  Nulled when the workload or the kernel sources differ from the profiled ones."""
    info = {"valu_bound_ms": None, "frac_of_valu_bound": None, "valu_bound_profile": "profiles/r06_tpi67_valu_bound.json"}
    try:
        with open(os.path.join(REPO, "profiles", "r06_tpi67_valu_bound.json")) as fh:
            prof = json.load(fh)
    except (OSError, ValueError):
        info["valu_bound_note"] = "no committed profile"
        return info
    if (ny, nx, size, world) != (32768, 32768, 67, 1):
        info["valu_bound_note"] = "profiled workload is 32768x32768, 67 px, 1 GPU"
    elif prof.get("kernel_sources_sha256") != kernel_sources_sha256():
        info["valu_bound_note"] = "kernel sources changed since tools/valu_bound.py ran"
    else:
        info["valu_bound_ms"] = prof.get("valu_bound_ms")
        info["frac_of_valu_bound"] = round(prof["valu_bound_ms"] / kernel_ms, 4) if prof.get("valu_bound_ms") else None
        info["valu_row_loop"] = prof.get("row_loop_instructions")
    return info


def stats(ms):
    ms = sorted(ms)
    n = len(ms)
    med = ms[n // 2] if n % 2 else 0.5 * (ms[n // 2 - 1] + ms[n // 2])
    return {"median": med, "mean": sum(ms) / n, "min": ms[0], "max": ms[-1], "n": n}


def time_kernel(fn, reps, dev):
    """Per-launch HIP-event durations of `reps` back-to-back launches of `fn` (after one untimed launch)."""
    return stats(dev.time_launches(fn, reps, warm=1))


def cpu_baseline(size, sample_rows, sample_cols, dem_sample):
    from oracle import topo_oracle as orc

    t0 = time.perf_counter()
    orc.tpi_scipy(dem_sample, size)
    dt = time.perf_counter() - t0
    return {
        "value": round(sample_rows * sample_cols / dt / 1e6, 3),
        "unit": "Mpixels/s",
        "cores": 1,
        "kind": "port",
        "sample": f"oracle.tpi_scipy (scipy.signal.convolve FFT path of the reference) on a "
                  f"{sample_rows}x{sample_cols} window of the same DEM, size {size}, "
                  f"{dt:.1f} s wall",
    }


def cpu_twin_baseline(size, dem_sample):
    """The oracle's C/OpenMP twin (exact float64 evaluation; row prefix sums in thread-private tiles, pixel loop
    innermost and vectorised) on every host core: TPI at `size` and Sx (azimuth 0, radius 500 m; the distinct ray
    pixels once each, one atan per pixel) on a bounded window of the same DEM.  The thread pool is started on a
    small window first and the better of two runs counts (the first run of a process pays for thread start-up and
    for faulting in the float64 result plane)."""
    from oracle import c_twin, topo_oracle as orc

    rows, cols = dem_sample.shape
    c_twin.tpi_std(dem_sample[:512, :512], size, want_tpi=True, want_std=False)
    plane = np.zeros(dem_sample.shape, np.float64)  # the result plane, touched once: not part of the timing
    dt_tpi = None
    for _ in range(3):
        t0 = time.perf_counter()
        c_twin.tpi_std(dem_sample, size, want_tpi=True, want_std=False, out_tpi=plane)
        dt = time.perf_counter() - t0
        dt_tpi = dt if dt_tpi is None else min(dt_tpi, dt)
    del plane
    window, offs, dist = orc.sx_geometry(0.0, 500.0, 30.0, -30.0)
    sx_sample = np.ascontiguousarray(dem_sample[:8192, :8192])  # the pool is warm by now
    t0 = time.perf_counter()
    c_twin.sx(sx_sample, offs[:, 0], offs[:, 1], dist, window, 10.0)
    dt_sx = time.perf_counter() - t0
    return {
        "kind": "port", "cores": c_twin.threads(), "unit": "Mpixels/s",
        "value": round(rows * cols / dt_tpi / 1e6, 2),
        "sx_az0_r500_value": round(sx_sample.shape[0] * sx_sample.shape[1] / dt_sx / 1e6, 2),
        "sample": f"oracle/topo_oracle.c (OpenMP, float64, warm thread pool) on windows of the same DEM: TPI size {size} on "
                  f"{rows}x{cols} in {dt_tpi:.2f} s (best of three runs into one result plane), Sx az 0 r 500 m on "
                  f"{sx_sample.shape[0]}x{sx_sample.shape[1]} in {dt_sx:.2f} s",
    }


# which kernels run behind each entry of `descriptors` (csrc/*.hip; the empty follow-up launches of the disc
# paths - fraction pass, general kernel over deferred tiles - are named in DESIGN.md section 3)
def disc_kernels(what, size):
    if what == "tpi":
        return ("tpi_ring_kernel<%d, 8, .>" % size if 5 <= size <= 11 else "tpi_march_kernel<%d, 60, 12, ..>" % size) + \
            " (+ fraction pass and general kernel over marked tiles: none on whole metres)"
    if 5 <= size <= 7 or (what == "tpi_std" and size <= 13):
        return "std_ring_spec_kernel<%d, %s, false, 8> (u and u^2 rings, one staging pass; 512-column strips: eight staging waves beside " \
               "eight chain waves of 8 columns per lane, one barrier per phase; the DEM's border tiles included) + disc_wave_kernel over " \
               "the marked tiles (none here)" % (size, "true" if what == "tpi_std" else "false")
    if 5 <= size <= 41:
        return "std_ring_spec_kernel<%d, %s> (u and u^2 rings, one staging pass; four staging waves beside eight chain waves, one " \
               "barrier per 16 rows; the DEM's border tiles included: padding staged as samples of elevation 0) + disc_wave_kernel " \
               "over the marked tiles (windows with more than 2 lim32 of relief: none here)" % (size, "true" if what == "tpi_std" else "false")
    if size <= 67:
        return "std_ring_kernel<%d, %s> (u and u^2 rings, one staging pass) + disc_wave_kernel over the marked tiles " \
               "(DEM border, windows with more than 2 lim32 of relief)" % (size, "true" if what == "tpi_std" else "false")
    return "tpi_march_kernel<%d, 60, 12, OUT_SUM> + std_march_kernel<%d, 60, 12> + disc_wave_kernel over marked tiles" % (size, size)


def extras(dev, lib_mod, args, block_cls, dem, ny, nx):
    """Per-descriptor throughput on the resident DEM (N = 1 only): not the headline.  Every entry is the
    median of >= 10 launches, each bracketed by its own HIP events (min / max / n alongside)."""
    from topo_descriptors_amd import device as d

    out = {}
    o1 = d.DeviceArray(ny, nx)
    o2 = d.DeviceArray(ny, nx)
    blk = block_cls(dem)
    px = ny * nx
    REPS = 10

    def entry(key, st, bpp, kernel, per=1):
        ms = st["median"] / per
        out[key] = {"ms": round(ms, 4), "ms_min": round(st["min"] / per, 4), "ms_max": round(st["max"] / per, 4),
                    "launches": st["n"], "Mpixels_per_s": round(px / ms / 1e3, 1),
                    "hbm_frac": round(px * bpp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "kernel": kernel}

    for size in (7, 65, 67):
        entry(f"tpi_s{size}", time_kernel(lambda: blk.tpi_std(size, tpi=o1), REPS, d), 8, disc_kernels("tpi", size))
        entry(f"std_s{size}", time_kernel(lambda: blk.tpi_std(size, std=o2), REPS, d), 8, disc_kernels("std", size))
        if size == 67 and (ny, nx) == (32768, 32768):
            # what bounds STD in practice: vector-ALU issue too (tools/valu_bound.py std: the launch's SQ_INSTS_VALU priced at
            # the phase loop's mix of issue classes)
            try:
                with open(os.path.join(REPO, "profiles", "r06_std67_valu_bound.json")) as fh:
                    prof = json.load(fh)
                if prof.get("kernel_sources_sha256") == kernel_sources_sha256() and prof.get("valu_bound_ms"):
                    out["std_s67"]["valu_bound_ms"] = prof["valu_bound_ms"]
                    out["std_s67"]["frac_of_valu_bound"] = round(prof["valu_bound_ms"] / out["std_s67"]["ms"], 4)
                    out["std_s67"]["valu_bound_profile"] = "profiles/r06_std67_valu_bound.json"
                else:
                    out["std_s67"]["valu_bound_note"] = "kernel sources changed since tools/valu_bound.py std ran"
            except (OSError, ValueError):
                out["std_s67"]["valu_bound_note"] = "no committed profile"
        entry(f"tpi_std_s{size}", time_kernel(lambda: blk.tpi_std(size, tpi=o1, std=o2), REPS, d), 12,
              disc_kernels("tpi_std", size))
    o3 = d.DeviceArray(ny, nx)
    o4 = d.DeviceArray(ny, nx)
    chunks = "8 row chunks, gradient_epilogue4_kernel of chunk k on a second stream beside the smooth of chunk k + 1"
    grad_kernels = {
        3.25: "gauss_fused_f16_kernel (both passes of the radius-13 filter in one kernel on the f16 matrix pipe, the "
              "intermediate plane in LDS; the two-pass kernels queued behind it return at once on a DEM without "
              "non-finite samples); " + chunks,
        30.25: "gauss_axis0_f16_kernel<18, 3, 2> (banded Toeplitz products as three v_mfma_f32_32x32x16_f16 per 16 taps, "
               "samples split into two f16 parts around the tile offset) + gauss_axis1_s1_kernel<9, 3> (samples split once, "
               "against the reference of their 64-column slab; 16-row bands, two waves per SIMD, v_mfma_f32_16x16x32_f16) + "
               "their repair passes; " + chunks}
    for sigma in (3.25, 30.25):
        # the Gaussian primitive (topo.dem) at the two sigmas: 4 B read + 4 B written per pixel
        entry(f"dem_sigma{sigma}", time_kernel(lambda: blk.gaussian(sigma, sigma, o1), REPS, d), 8,
              grad_kernels[sigma].split(";")[0].replace(" + their repair passes", "") + " (no row chunks)")
        fn = lambda: blk.gradient(sigma, [30.0], [-30.0], dx=o1, dy=o2, slope=o3, aspect=o4)  # noqa: E731
        entry(f"gradient_sigma{sigma}", time_kernel(fn, REPS, d), 20, grad_kernels[sigma])
        fn = lambda: blk.gradient(sigma, [30.0], [-30.0], slope=o3, aspect=o4)  # noqa: E731
        entry(f"slope_aspect_sigma{sigma}", time_kernel(fn, REPS, d), 12, grad_kernels[sigma])
    # azimuth 45: the weak case (diagonal sectors cut into short chains)
    for azimuth, radius in ((0.0, 500.0), (0.0, 2000.0), (45.0, 500.0), (45.0, 2000.0)):
        window, dj, di, dist = d.sx_offsets(azimuth, radius, 30.0, -30.0)
        fn = lambda: blk.sx(dj, di, dist, window, 10.0, o1)  # noqa: E731
        entry(f"sx_az{int(azimuth)}_r{int(radius)}", time_kernel(fn, REPS, d), 8, "sx_kernel<stride> (LDS tile, chains of 8 / 4 / 2 neighbouring ray pixels down the columns, along the rows or along a diagonal, whichever needs the fewest comparisons; chains with equal weights - the two sides of a sector - scanned as one; one atan per pixel)")
    # two small discs in one pass over the DEM: ms for the pair, rate and fraction per plane
    st = time_kernel(lambda: blk.tpi_multi([7, 11], [o1, o2]), REPS, d)
    entry("tpi_s7_s11_one_pass_per_plane", st, 6, "tpi_ring_kernel<11, 8, kRingMain, 7> (one staging pass, one ring, two chains; 4 B read + 8 B "
          "written per pixel = 6 B per plane)", per=2)
    # 8 azimuths every 5 degrees in one pass: ms and rate are per azimuth plane
    sectors = [d.sx_offsets(5.0 * k, 500.0, 30.0, -30.0) for k in range(8)]
    fan = [o1, o2, o3, o4] + [d.DeviceArray(ny, nx) for _ in range(4)]
    entry("sx_r500_8_azimuths_step5_per_azimuth", time_kernel(lambda: blk.sx_multi(sectors, 10.0, fan), REPS, d), 8,
          "sx_multi_kernel<8>", per=8)
    for a in fan[4:]:
        a.free()
    # valley index at 200 m (7 px): 180 angles x 3 plane sums of rotated kernels in one pass.  Round 6: a dense
    # product on the matrix pipe over the 53 cells of the 10 x 10 canvas that hold a tap at any angle (csrc/valley_mfma.hip);
    # the tap-by-tap float32 kernel of rounds 2 - 5 (csrc/valley.hip) is timed beside it
    from topo_descriptors_amd import topo
    mean, stdev = d.mean_std(dem)
    taps, ksize, angles = topo._valley_ridge_tables(topo._valley_kernels(7, [0, 0.15, 0.3]),
                                                    np.arange(0, 180, dtype=np.float32))
    st = time_kernel(lambda: blk.valley_ridge(taps, ksize, angles, 3, mean, stdev, o1, o2), REPS, d)
    route = d.valley_route()
    # what the matrix pipe executes: cells (or, folded, pairs of cells) that hold a tap at any angle -> K steps of 16; filter tiles
    # of 10 angles x 3 planes, per class of canvas centre when folded, the last group filled up; 3 products
    kmax = int(ksize.max())
    cells, members = {}, {}
    pos = 0
    for ks in ksize:
        sh = kmax // 2 - ks // 2
        centre = int(ks) - 1 + 2 * sh if route & 8 else -1
        live = np.any(taps[pos:pos + ks * ks * 4].reshape(ks, ks, 4)[:, :, :3] != 0, axis=2)
        for ky, kx in zip(*np.nonzero(live)):
            cell = (ky + sh) * kmax + kx + sh
            cells.setdefault(centre, set()).add(min(cell, (centre - ky - sh) * kmax + centre - kx - sh) if route & 8 else cell)
        members[centre] = members.get(centre, 0) + 1
        pos += ks * ks * 4
    ksteps = max((len(c) + 15) // 16 for c in cells.values())
    group = 4 if ksteps <= 3 else 2 if ksteps <= 7 else 1
    tiles = sum(-(-n // 10) for n in members.values()) if route & 8 else sum(-(-(-(-n // 10)) // group) * group for n in members.values())
    mfmas = (px / 32) * tiles * ksteps * 3          # per launch; 32 x 32 x 16 multiply-adds each
    label = (f"valley_fold_kernel<{ksteps}, 3> (v_mfma_f32_32x32x16_f16 over PAIRS of window cells - the kernels are point-symmetric: "
             f"{' + '.join(str(len(c)) for c in cells.values())} live pairs in the two classes of canvas centre, {ksteps} K steps, {tiles} "
             "filter tiles of 10 angles x 3 planes, split-f16 operands: 3 products)" if route & 8 else
             f"valley_mfma_kernel<{ksteps}, 3> (v_mfma_f32_32x32x16_f16 over the {sum(len(c) for c in cells.values())} window cells that hold "
             f"a tap, {ksteps} K steps, {tiles} filter tiles, split-f16 operands: 3 products)")
    entry("valley_ridge_s7", st, 12, label + " + valley_ridge_kernel<3> over the tiles with non-finite samples (none here)"
          if route & 1 else "valley_ridge_kernel<3>")
    out["valley_ridge_s7"]["route"] = route
    out["valley_ridge_s7"]["k_steps"] = ksteps
    out["valley_ridge_s7"]["filter_tiles"] = tiles
    out["valley_ridge_s7"]["mfma_TFLOP_per_s_executed"] = round(mfmas * 32768 / st["median"] / 1e9, 1)
    out["valley_ridge_s7"]["mfma_frac_of_2500_TFLOP_per_s"] = round(mfmas * 32768 / st["median"] / 1e9 / 2500.0, 3)
    os.environ["TOPO_AMD_VALLEY_MFMA_MAX_KERNEL"] = "0"   # read at every launch
    st = time_kernel(lambda: blk.valley_ridge(taps, ksize, angles, 3, mean, stdev, o1, o2), max(2, REPS // 3), d)
    del os.environ["TOPO_AMD_VALLEY_MFMA_MAX_KERNEL"]
    entry("valley_ridge_s7_tap_by_tap", st, 12, "valley_ridge_kernel<3> (the float32 chain on the non-zero taps: rounds 2 - 5)")
    nonzero = int(np.any(taps.reshape(-1, 4)[:, :3] != 0, axis=1).sum())  # the taps the kernel evaluates
    out["valley_ridge_s7_tap_by_tap"]["GFMA_per_s_executed"] = round(px * nonzero * 3 / st["median"] / 1e6, 0)
    # a mid-size kernel (21 px: rotated kernels of 30 cells a side, ~290 pairs of cells a class): the streamed form of the matrix-pipe
    # kernel (round 6; the tap-by-tap kernel takes 358 ms for this on 8192^2, i.e. ~5.7 s here)
    taps21, ksize21, angles21 = topo._valley_ridge_tables(topo._valley_kernels(21, [0, 0.15, 0.3]), np.arange(0, 180, dtype=np.float32))
    st = time_kernel(lambda: blk.valley_ridge(taps21, ksize21, angles21, 3, mean, stdev, o1, o2), 3, d)
    entry("valley_ridge_s21", st, 12, "valley_fold_stream_kernel<3> (the folded product with the pixel operands streamed in chunks of 2 K steps, "
          "3 filter tiles x 2 pixel tiles of accumulators a wave)" if d.valley_route() & 16 else "valley_ridge_kernel<3>")
    out["valley_ridge_s21"]["route"] = d.valley_route()

    # the same TPI on fractional elevations: every tile runs the integer pass plus the float
    # chain on the fractional parts and goes through the per-row scratch planes (two passes)
    frac = d.synth_dem(ny, nx, seed=0, integer=False)
    fblk = block_cls(frac)
    for size in (7, 67):
        entry(f"tpi_s{size}_fractional_dem", time_kernel(lambda: fblk.tpi_std(size, tpi=o1), REPS, d), 8,
              "tpi_march_kernel (sums of trunc x) + tpi_fraction_march_kernel" if size > 17 else
              "tpi_ring_kernel<., kRingMark> (whole-metre tiles, marks) + tpi_ring_kernel<., kRingBoth> (second image: the "
              "fractional parts in 2^-16 m)")
        entry(f"std_s{size}_fractional_dem", time_kernel(lambda: fblk.tpi_std(size, std=o2), REPS, d), 8,
              "std_ring_spec_kernel<.> (probe: hands the runs over) + std_ring_spec_kernel<., ., true> (second pass with a third image: the fractional parts)" if size <= 21 else
              "tpi_march_sums_kernel + std_march_kernel<FRAC_STORE> + tpi_fraction_march_kernel<WANT_STD> (three marching passes)")
    frac.free()
    for a in (o1, o2, o3, o4):
        a.free()
    return out


def config_rows(d):
    """BASELINE.json configs[1..3] at their OWN sizes (the `descriptors` table above times everything at the headline's
    32768^2): 8192^2 TPI / STD at 7 and 65 px, 16384^2 gradient at sigma 3.25 and 30.25, 16384^2 Sx azimuth 0 radius
    500 m.  Median of 10 launches, HIP events, DEM resident."""
    out = {}

    def entry(key, n, st, bpp):
        ms = st["median"]
        out[key] = {"dem": [n, n], "ms": round(ms, 4), "ms_min": round(st["min"], 4), "ms_max": round(st["max"], 4),
                    "launches": st["n"], "Mpixels_per_s": round(n * n / ms / 1e3, 1),
                    "hbm_frac": round(n * n * bpp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    n = 8192
    dem = d.synth_dem(n, n, seed=0)
    outs = [d.DeviceArray(n, n) for _ in range(2)]
    blk = d.Block(dem)
    for size in (7, 65):
        entry(f"config2_8192_tpi_s{size}", n, time_kernel(lambda: blk.tpi_std(size, tpi=outs[0]), 10, d), 8)
        entry(f"config2_8192_std_s{size}", n, time_kernel(lambda: blk.tpi_std(size, std=outs[1]), 10, d), 8)
        entry(f"config2_8192_tpi_std_s{size}", n, time_kernel(lambda: blk.tpi_std(size, tpi=outs[0], std=outs[1]), 10, d), 12)
    for a in outs + [dem]:
        a.free()
    n = 16384
    dem = d.synth_dem(n, n, seed=0)
    outs = [d.DeviceArray(n, n) for _ in range(4)]
    blk = d.Block(dem)
    for sigma in (3.25, 30.25):
        fn = lambda: blk.gradient(sigma, [30.0], [-30.0], dx=outs[0], dy=outs[1], slope=outs[2], aspect=outs[3])  # noqa: E731
        entry(f"config3_16384_gradient_sigma{sigma}", n, time_kernel(fn, 10, d), 20)
        fn = lambda: blk.gradient(sigma, [30.0], [-30.0], slope=outs[2], aspect=outs[3])  # noqa: E731
        entry(f"config3_16384_slope_aspect_sigma{sigma}", n, time_kernel(fn, 10, d), 12)
    window, dj, di, dist = d.sx_offsets(0.0, 500.0, 30.0, -30.0)
    entry("config4_16384_sx_az0_r500", n, time_kernel(lambda: blk.sx(dj, di, dist, window, 10.0, outs[0]), 10, d), 8)
    for a in outs + [dem]:
        a.free()
    return out


def end_to_end(d, lib_mod, size, n=16384):
    """topo.tpi through the HOST-buffer entry point (upload + kernels + download, what `topo.tpi(numpy array)` costs) on an
    n x n DEM: pageable numpy arrays (second call: the result array's pages are recycled) and page-locked ones
    (topo_amd_host_alloc), next to the three phases timed one by one on device-resident buffers.  Never the bench
    `value` (reported separately)."""
    import ctypes as C

    from topo_descriptors_amd import topo
    lib = lib_mod.lib()
    dem = np.rint(1900.0 + 300.0 * np.random.default_rng(0).standard_normal((n, n))).astype(np.float32)
    out = {"dem": [n, n], "disc_px": size, "bytes_moved": 2 * dem.nbytes}
    warm = topo.tpi(dem, size)
    t0 = time.perf_counter()
    res = topo.tpi(dem, size)
    dt = time.perf_counter() - t0
    del warm
    out["pageable_ms"] = round(dt * 1e3, 2)
    out["pageable_Mpixels_per_s"] = round(n * n / dt / 1e6, 1)
    # page-locked arrays
    hin, hout = C.c_void_p(), C.c_void_p()
    lib_mod.check(lib.topo_amd_host_alloc(C.byref(hin), dem.nbytes), "host_alloc")
    lib_mod.check(lib.topo_amd_host_alloc(C.byref(hout), dem.nbytes), "host_alloc")
    pin_in = np.frombuffer((C.c_char * dem.nbytes).from_address(hin.value), dtype=np.float32).reshape(n, n)
    pin_out = np.frombuffer((C.c_char * dem.nbytes).from_address(hout.value), dtype=np.float32).reshape(n, n)
    pin_in[:] = dem
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        lib_mod.check(lib.topo_amd_tpi_f32(hin, n, n, int(size), 0.0, hout), "topo_amd_tpi_f32")
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["pinned_ms"] = round(best * 1e3, 2)
    out["pinned_Mpixels_per_s"] = round(n * n / best / 1e6, 1)
    out["pinned_equals_pageable"] = bool(np.array_equal(pin_out, res))
    del res
    # the phases, device-resident: H2D, kernels (HIP events), D2H
    dev, o = d.DeviceArray(n, n), d.DeviceArray(n, n)
    t0 = time.perf_counter()
    lib_mod.check(lib.topo_amd_memcpy_h2d(dev.ptr, hin, dem.nbytes), "h2d")
    d.sync()
    out["h2d_pinned_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    blk = d.Block(dev)
    out["kernel_ms"] = round(time_kernel(lambda: blk.tpi_std(size, tpi=o), 5, d)["median"], 4)
    t0 = time.perf_counter()
    lib_mod.check(lib.topo_amd_memcpy_d2h(hout, o.ptr, dem.nbytes), "d2h")
    out["d2h_pinned_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    dev.free()
    o.free()
    del pin_in, pin_out
    lib_mod.check(lib.topo_amd_host_free(hin), "host_free")
    lib_mod.check(lib.topo_amd_host_free(hout), "host_free")
    return out


def parity_spots(d, block, first_row, nx):
    """Spot parity at full size, beyond TPI: STD, slope / aspect and Sx of windows of the bench DEM against the oracle's
    restatement of the reference's scipy / numba calls (interiors only: the windows are cut out of a larger DEM)."""
    from oracle import c_twin, topo_oracle as orc

    out = {}
    n = 4096
    sample = block.to_host(first_row, n)[:, :n].copy()
    dev = d.DeviceArray.from_host(sample)
    blk = d.Block(dev)
    o = [d.DeviceArray(n, n) for _ in range(4)]
    for size in (7, 67):
        blk.tpi_std(size, std=o[0])
        d.sync()
        got = o[0].to_host().astype(np.float64)
        want = c_twin.tpi_std(sample, size, want_tpi=False, want_std=True)[1]  # exact float64 evaluation (oracle/topo_oracle.c)
        ref = orc.std_scipy(sample, size)                                       # the reference's two FFT convolutions
        r = size
        inner = (slice(r, n - r), slice(r, n - r))
        out[f"std_s{size}"] = {"max_abs_err_vs_exact": float(np.max(np.abs(got[inner] - want[inner]))),
                               "reference_floor_vs_exact": float(np.max(np.abs(ref[inner] - want[inner]))),
                               "max_std": float(np.max(want[inner])), "window": [n - 2 * r, n - 2 * r]}
    res = {"x": 30.0, "y": -30.0}
    for sigma in (3.25, 30.25):
        blk.gradient(sigma, [30.0], [-30.0], dx=o[0], dy=o[1], slope=o[2], aspect=o[3])
        d.sync()
        got = [a.to_host() for a in o]
        want = orc.gradient_scipy(sample, sigma, res)
        r = int(4 * sigma + 0.5) + 2
        inner = (slice(r, n - r), slice(r, n - r))
        steep = want[2][inner] > 0.1
        out[f"gradient_sigma{sigma}"] = {
            "slope_max_abs_err_deg": float(np.max(np.abs(got[2][inner] - want[2][inner]))),
            "slope_rel_range": float(np.max(np.abs(got[2][inner] - want[2][inner])) / np.max(np.abs(want[2][inner]))),
            "aspect_max_wrapped_err_deg_where_slope_gt_0.1": float(np.max(orc.wrapped_angle_diff(got[3][inner], want[3][inner])[steep])),
            "dx_rel_range": float(np.max(np.abs(got[0][inner] - want[0][inner])) / np.max(np.abs(want[0][inner]))),
            "window": [n - 2 * r, n - 2 * r]}
        # the Gaussian primitive itself against the float64 evaluation of the same filter (whole sample: reflect at its edges)
        blk.gaussian(sigma, sigma, o[0])
        d.sync()
        out[f"dem_sigma{sigma}"] = {"max_abs_err_m_vs_float64": float(np.max(np.abs(o[0].to_host().astype(np.float64) - orc.gaussian_exact(sample, sigma)))),
                                    "window": [n, n]}
    window, dj, di, dist = d.sx_offsets(0.0, 500.0, 30.0, -30.0)
    blk.sx(dj, di, dist, window, 10.0, o[0])
    d.sync()
    got = o[0].to_host()
    w2, offs, dist2 = orc.sx_geometry(0.0, 500.0, 30.0, -30.0)
    want = c_twin.sx(sample, offs[:, 0], offs[:, 1], dist2, w2, 10.0)
    inner = (slice(window, n - window), slice(window, n - window))
    out["sx_az0_r500"] = {"max_abs_err_deg": float(np.max(np.abs(got[inner] - want[inner]))),
                          "checker": "oracle/topo_oracle.c (float64 twin of _sx_rolling)", "window": [n - 2 * window, n - 2 * window]}
    for a in o + [dev]:
        a.free()
    return out


SHARD_KEYS = ("tpi_s67", "std_s67", "tpi_std_s67", "gradient_sigma3.25", "gradient_sigma30.25", "sx_az0_r500")
SHARD_BYTES = {"tpi_s67": 8, "std_s67": 8, "tpi_std_s67": 12, "gradient_sigma3.25": 20, "gradient_sigma30.25": 20,
               "sx_az0_r500": 8}


def sharded_steps(sd, d, rows_local, nx, outs=None):
    """One collective step per descriptor of BASELINE configs[4] on this rank's shard: {key: (callable, outputs)}.
    Every call refreshes the descriptor's own ghost rows over RCCL (overlapped with the interior rows) and computes
    the seam rows behind the ghost-row gate.  (`sd` may also be a device.Block: the same calls on one block.)"""
    outs = outs or [d.DeviceArray(rows_local, nx) for _ in range(4)]
    window, dj, di, dist = d.sx_offsets(0.0, 500.0, 30.0, -30.0)
    return outs, {
        "tpi_s67": lambda: sd.tpi_std(67, tpi=outs[0]),
        "std_s67": lambda: sd.tpi_std(67, std=outs[1]),
        "tpi_std_s67": lambda: sd.tpi_std(67, tpi=outs[0], std=outs[1]),
        "gradient_sigma3.25": lambda: sd.gradient(3.25, [30.0], [-30.0], dx=outs[0], dy=outs[1], slope=outs[2],
                                                  aspect=outs[3]),
        "gradient_sigma30.25": lambda: sd.gradient(30.25, [30.0], [-30.0], dx=outs[0], dy=outs[1], slope=outs[2],
                                                   aspect=outs[3]),
        "sx_az0_r500": lambda: sd.sx(dj, di, dist, window, 10.0, outs[0]),
    }


def sharded_descriptors(rdv, steps, time_launches, px_total, world, reps=10):
    """Per-descriptor table of the row-sharded run: every rank times its own launches with HIP events, the slowest
    rank's median counts (a step is over when the last shard is), and the rate is the whole DEM over that time.
    Collective: every rank calls it with the same keys in the same order."""
    out = {}
    for key in SHARD_KEYS:
        st = stats(time_launches(steps[key], reps, 2))
        ms = rdv.max(st["median"])
        ms_min, ms_max = -rdv.max(-st["min"]), rdv.max(st["max"])
        bpp = SHARD_BYTES[key]
        out[key] = {"ms": round(ms, 4), "ms_min": round(ms_min, 4), "ms_max": round(ms_max, 4), "launches": st["n"],
                    "Mpixels_per_s": round(px_total / ms / 1e3, 1),
                    # fraction of the aggregate roofline of the GPUs that took part
                    "hbm_frac": round(px_total * bpp / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4),
                    "entry_point": "topo_amd_shard_" + ("tpi_std" if "s67" in key else
                                                         "gradient" if key.startswith("gradient") else "sx")}
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched with torch.distributed.run "
                     "(one rank per GPU)")
        args.gpus = world
    rdv = Rendezvous(world)
    rank = rdv.rank
    os.environ.setdefault("TOPO_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0"))

    import ctypes

    from topo_descriptors_amd import _lib, device as d

    lib = _lib.lib()  # binds this process to its GPU; raises when the HIP library is missing
    ny, nx, size = args.ny, args.nx, args.size

    # ---- row shard of this rank ---------------------------------------------------------------
    base, extra = divmod(ny, world)
    rows_local = base + (1 if rank < extra else 0)
    row0 = rank * base + min(rank, extra)
    up, down = ctypes.c_int32(), ctypes.c_int32()
    _lib.check(lib.topo_amd_halo_rows(_lib.DESC_TPI, float(size), 0.0, ctypes.byref(up),
                                      ctypes.byref(down)), "halo_rows")
    loopback = world == 1 and os.environ.get("TOPO_AMD_HALO_LOOPBACK") == "1"
    sharded = world > 1 or loopback
    sd = None
    if sharded:
        from topo_descriptors_amd import shard as shard_mod

        shard_mod.ShardedDEM.init_comm(rank, world, rdv.bcast_bytes)
        # one buffer for every descriptor of the table: ghost zones as deep as the deepest asks for (the gradient
        # at sigma 30.25: 122 rows); each call uses the rows next to the owned ones (topo_amd_shard_layout)
        deep = max(max(shard_mod.halo_rows(_lib.DESC_TPI, size)),
                   0 if args.no_extras else max(shard_mod.halo_rows(_lib.DESC_GRADIENT, 30.25, 1.0)))
        if loopback:  # the middle shard of three; its neighbours are itself
            plan = shard_mod.RowShardPlan(3 * ny, nx, 3, 1, deep, deep)
        else:
            plan = shard_mod.RowShardPlan(ny, nx, world, rank, deep, deep)
        assert loopback or (plan.row0, plan.rows_local) == (row0, rows_local)
        sd = shard_mod.ShardedDEM(plan)
        block = sd.block
        halo_up, halo_dn = up.value, down.value
        first_row = plan.halo_above
    else:
        halo_up = halo_dn = first_row = 0
        block = d.DeviceArray(rows_local, nx)
    d.synth_dem(rows_local, nx, row0=row0, seed=0, out=block, out_row=first_row)
    out = d.DeviceArray(rows_local, nx)
    d.sync()
    # (sharded: the first topo_amd_shard_* call on the freshly written shard classifies the whole raster itself - collective,
    # in the warm-up - so every shard takes the kernels the single GPU takes)

    if not sharded:
        blk = d.Block(block)

        def step():
            blk.tpi_std(size, tpi=out)
    else:
        def step():
            sd.tpi_std(size, tpi=out)

    # bring the clocks up: untimed launches until three consecutive ones agree within 1.5 % (at most 40).
    # Every rank runs the same number (the count is agreed on through the max over ranks).
    ramp, last = 0, []
    while ramp < 40:
        d.mark(0)
        step()
        d.mark(1)
        last = (last + [d.mark_elapsed(0, 1)])[-3:]
        ramp += 1
        settled = len(last) == 3 and max(last) <= 1.015 * min(last) and ramp >= 4
        if rdv.max(0.0 if settled else 1.0) == 0.0:
            break
    for _ in range(args.warmup):
        step()
    d.sync()
    rdv.barrier()
    steps = args.steps
    if steps > 510:
        sys.exit("bench.py: at most 510 timed steps (one HIP event per step)")
    t0 = time.perf_counter()
    d.mark(0)
    for k in range(steps):
        step()
        d.mark(k + 1)  # HIP events on the compute stream between the K launches, no host synchronise
    d.sync()
    rdv.barrier()
    wall = time.perf_counter() - t0
    wall = rdv.max(wall)
    per_step = stats([d.mark_elapsed(k, k + 1) for k in range(steps)])
    kernel_ms = rdv.max(per_step["mean"])
    kernel_ms_median = rdv.max(per_step["median"])

    px_total = ny * nx
    value = px_total * args.steps / wall / 1e6
    result = None
    if rank == 0:
        px_launch = rows_local * nx  # what one rank's launch covers
        achieved = px_launch * BYTES_PER_PIXEL["tpi"] / (kernel_ms * 1e-3) / 1e9
        result = {
            "metric": "Mpixels/s (and % HBM roofline) per descriptor, 1/2/4/8 MI355X",
            "value": round(value, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4),
            "ms_per_step_median": round(kernel_ms_median, 4),
            "ms_per_step_mean": round(kernel_ms, 4),
            "ms_per_step_min": round(per_step["min"], 4),
            "ms_per_step_max": round(per_step["max"], 4),
            "value_at_median": round(px_total / kernel_ms_median / 1e3, 1),
            "clock_ramp_steps": ramp,
            "higher_is_better": True,
            "scaling": "strong",
            "exchange": ("none (one block)" if not sharded else
                         "loop-back: ncclSend / ncclRecv of the ghost rows to this same GPU (both ends of every link "
                         "are one device: the exchange is exercised, not a scaling number)" if loopback else
                         "ncclSend / ncclRecv of the ghost rows to the neighbour ranks over RCCL, overlapped with "
                         "the interior rows"),
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"topo.tpi size={size}px (2000 m @ 30 m) on a {ny}x{nx} f32 DEM with "
                            f"integer-valued elevations (whole metres: every tile takes the one-pass "
                            f"TPI path; see descriptors.tpi_s{size}_fractional_dem for the two-pass "
                            f"rate on fractional elevations), row-sharded over {world} GPU(s)",
                "descriptor": "tpi", "disc_px": size, "dem": [ny, nx],
                "rows_per_gpu": rows_local, "halo_rows": [halo_up, halo_dn],
                "parallelism": f"rows{world}",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "frac_at_median": round(px_launch * BYTES_PER_PIXEL["tpi"] / (kernel_ms_median * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                # north_star's literal "HBM-read roofline": the 4 B/pixel the kernel has to read, nothing else
                "frac_read_only_basis": round(px_launch * 4 / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                **measured_traffic(ny, nx, size, 0 if loopback else world),
                "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes)",
                "algorithmic_bytes_per_launch": px_launch * BYTES_PER_PIXEL["tpi"],
                "kernel": "tpi_march_kernel<67, 60, 12, true, true, true> (exact one-pass TPI on whole-metre tiles, marching down "
                          "column strips), per-rank launch; kernel_ms also covers the general kernel launched after "
                          "it over the tiles it deferred and the fraction pass for tiles with fractional elevations "
                          "(neither finds a tile on this DEM, ~10 us together)",
                "kernel_ms": round(kernel_ms, 4),
                "kernel_ms_median": round(kernel_ms_median, 4),
                # what bounds it (profiles/r02_valu_mix_rate.txt, r03_valu_mix2_rate.txt, r03_tpi67_valu_bound.json,
                # DESIGN.md K1): the exact disc sum is vector-ALU work - 315 vector instructions per wave and output row,
                # 171 of them at half rate (every DPP form, v_add3_u32, converts, float64) - i.e. ~3.1 ms per launch at
                # 100 % VALU utilisation; valu_bound_ms / frac_of_valu_bound below
                "bound_in_practice": "valu",
                # the >= 60 % of north_star is out of reach of an exact disc sum - the kernel's own instruction
                # stream, priced at measured issue rates, takes 3.0 ms of vector-ALU time per launch (frac 0.35)
                "target_60pct": "not reachable with an exact disc: VALU bound 0.35",
                **valu_bound(ny, nx, size, 0 if loopback else world, kernel_ms),
                "algorithmic_bytes_per_pixel": BYTES_PER_PIXEL["tpi"],
            },
        }
        if not sharded:
            # what the memory system gives a kernel that does NOTHING but move the same 8 B/pixel (round 6): the runtime's
            # device-to-device copy of the DEM plane into the output plane, timed like the kernel (HIP events on the compute
            # stream, behind the timed region).  tools/ubench/strip_copy.hip measures the same with the kernels' own access
            # pattern - 128-column strips marched down: 1.54 ms = 5.6 TB/s (profiles/r06_gauss_axis0_s1.txt, 3b).
            try:
                lib = _lib.lib()
                nbytes = rows_local * nx * 4
                copies = []
                for k in range(4):
                    d.mark(0)
                    _lib.check(lib.topo_amd_memcpy_d2d(out.ptr, block.row_ptr(first_row), nbytes), "memcpy_d2d")
                    d.mark(1)
                    copies.append(d.mark_elapsed(0, 1))
                copy_ms = sorted(copies[1:])[1]
                result["roofline"]["hbm_copy_live"] = {
                    "what": "hipMemcpyAsync device-to-device of one plane (4 B/pixel read + 4 B/pixel written, like the kernel)",
                    "ms": round(copy_ms, 4), "GB/s": round(2 * nbytes / (copy_ms * 1e-3) / 1e9, 1),
                    "kernel_over_copy": round(kernel_ms / copy_ms, 3)}
                step()  # (the output plane holds the kernel's result again for the parity spot below)
                d.sync()
            except Exception as exc:  # noqa: BLE001
                result["roofline"]["hbm_copy_live"] = {"error": repr(exc)}
        if not args.no_cpu and not sharded:
            rows_s = min(ny, 16384)  # ~4 s of scipy on one core + ~6 s of the C twin on all of them
            cols_s = min(nx, 16384)
            sample = block.to_host(first_row, rows_s)[:, :cols_s].copy()
            result["cpu_baseline"] = cpu_baseline(size, rows_s, cols_s, sample)
            result["cpu_baseline_all_cores"] = cpu_twin_baseline(size, sample)
            # spot parity at full size: TPI of the same window vs the oracle, interior only
            got = out.to_host(0, rows_s)[:, :cols_s]
            from oracle import topo_oracle as orc

            r = size
            want = orc.tpi_scipy(sample, size)
            result["parity_spot"] = {
                "max_abs_err_m": float(np.max(np.abs(got[: rows_s - r, : cols_s - r] -
                                                      want[: rows_s - r, : cols_s - r]))),
                "window": [rows_s - r, cols_s - r],
                "checker": "oracle.tpi_scipy (the reference's scipy.signal.convolve call)",
            }
            result["parity_spot"].update(parity_spots(d, block, first_row, nx))
        if not args.no_extras and not sharded:
            result["descriptors"] = extras(d, _lib, args, d.Block, block, ny, nx)
            result["descriptors_at_config_sizes"] = config_rows(d)
            if not args.no_end_to_end:
                result["end_to_end"] = end_to_end(d, _lib, size)
    if not args.no_extras and sharded:  # collective: every rank takes part, rank 0 reports
        outs, steps_by_key = sharded_steps(sd, d, rows_local, nx)
        table = sharded_descriptors(rdv, steps_by_key, d.time_launches, px_total, world)
        d.sync()
        for a in outs:
            a.free()
        gave_up = ctypes.c_uint()
        _lib.check(lib.topo_amd_gate_giveups(ctypes.byref(gave_up)), "gate_giveups")
        gave_up_max = int(rdv.max(float(gave_up.value)))
        if loopback and rank == 0:
            # shard_efficiency = (single-GPU time of the DEM this shard is one eighth of) / (8 x the shard's step): what
            # 8-way strong scaling could reach at best, before any link cost (both ends of the exchange are this GPU)
            parts = 8
            full = d.synth_dem(parts * ny, nx, seed=0)
            fouts = [d.DeviceArray(parts * ny, nx) for _ in range(4)]
            fblk = d.Block(full)
            _, full_steps = sharded_steps(fblk, d, parts * ny, nx, outs=fouts)
            for key in SHARD_KEYS:
                full_ms = stats(d.time_launches(full_steps[key], 6, 2))["median"]
                table[key]["single_gpu_ms_whole_dem"] = round(full_ms, 4)
                table[key]["shard_efficiency"] = round(full_ms / (parts * table[key]["ms"]), 4)
            d.sync()
            for a in fouts + [full]:
                a.free()
        if rank == 0:
            result["descriptors"] = table
            result["gate_giveups"] = gave_up_max  # blocks that found the ghost-row gate closed (0: the exchange hid behind the interior rows)
            result["descriptors_note"] = (
                "row-sharded entry points (topo_amd_shard_*), one ghost-row exchange per step and descriptor; ms = "
                "median of 10 launches on the slowest rank; Mpixels_per_s = the whole DEM over that time; hbm_frac "
                "against n_gpus x 8 TB/s")
    out.free()
    block.free()
    if sharded:
        _lib.check(lib.topo_amd_comm_destroy(), "comm_destroy")
    rdv.close()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
