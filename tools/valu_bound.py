#!/usr/bin/env python3
"""Vector-ALU bound of the headline kernel (tpi_march_kernel<67, 60, 12>) from its own instruction stream.

What bounds TPI at 67 px is vector-ALU issue, not HBM.  This tool makes that a number the bench
line can carry: it compiles the kernel from the product's headers (tools/ubench/tpi_lab.hip, device code only),
finds the row loop in the ISA (the loop that holds the 42 ds_read_b128 of a 67-px chain), counts its vector
instructions by issue class and prices them with the issue costs measured on the GPU
(profiles/r02_valu_mix_rate.txt, profiles/r03_valu_mix2_rate.txt: 1.05 ns per wave-instruction and SIMD for plain
VOP1 / VOP2 integer and float operations, 1.87 ns for every DPP form, v_add3_u32, VOP3-only integer operations,
converts and float64).  The instructions outside the row loop (staging, tile bookkeeping) come from the launch's
SQ_INSTS_VALU counter minus the row loop's share and are priced at the kernel's overall mix.

    python tools/valu_bound.py [SQ_INSTS_VALU per launch, default from profiles/r06_tpi67_pmc_summary.txt] > profiles/r06_tpi67_valu_bound.json
    python tools/valu_bound.py std [SQ_INSTS_VALU per launch, default from profiles/r06_std67_pmc_summary.txt] > profiles/r06_std67_valu_bound.json

std (round 4): the same for std_ring_kernel<67, false> - its phase loop (one output row of 256 pixels per wave and
phase: two chains, the staging share of the wave, the finalisation) priced by issue class; the scalar instructions of
the loop are counted next to it (they issue beside the vector ones of the other waves of the SIMD).
"""
import json
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

NS_PLAIN, NS_HALF = 1.05, 1.87          # ns per wave-instruction and SIMD (16 waves per CU, wall clock)
NY = NX = 32768
SIZE, TH, TILE_W = 67, 60, 184
KERNEL = "tpi_march_kernelILi67ELi60ELi12ELb1ELb1ELb1"
STD_KERNEL = "std_ring_kernelILi67ELb0ELi0"
SUFFIX = "EEEvNS0_8WaveArgsEiiNS0_7PartRunE:"   # (WaveArgs, int, int, PartRun): the ordinary (one-part) kernels


def half_rate(op):
    return ("_dpp" in op or "add3" in op or "f64" in op or op.startswith("v_cvt") or op.startswith("v_mad") or
            op.startswith("v_mul_lo") or op.startswith("v_mul_hi") or op.startswith("v_lshl_add") or
            op.startswith("v_bfe") or op.startswith("v_perm") or op.startswith("v_cndmask") and op.endswith("e64") or
            op.startswith("v_cmp") or op.startswith("v_max3") or op.startswith("v_min3") or op.startswith("v_readlane") or
            op.startswith("v_writelane"))


def main():
    std = len(sys.argv) > 1 and sys.argv[1] == "std"
    if std:
        del sys.argv[1]
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "lab.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                        "--cuda-device-only", os.path.join(REPO, "tools", "ubench", "tpi_lab.hip"), "-o", asm],
                       check=True, stderr=subprocess.DEVNULL)
        txt = open(asm).read()
    start = txt.index((STD_KERNEL if std else KERNEL) + SUFFIX)
    lines = txt[start:txt.index("s_endpgm", start)].split("\n")
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))

    def ops(a, b):
        out = []
        for x in lines[a:b + 1]:
            x = x.strip()
            if x and not x.startswith((".", ";", "//")) and not x.endswith(":"):
                out.append(x.split()[0])
        return out

    # the row loop: the innermost loop with the chain's 42 prefix-row reads (+ 2 for the pixel's own value) that also
    # converts to float64 (the whole-metre copy of the loop; the sums-only copy for fractional tiles does not)
    best = None
    for a, b in loops:
        o = ops(a, b)
        reads = sum(1 for x in o if x == "ds_read_b128")
        if std:
            # the phase loop: the smallest loop that holds both chains' prefix-row reads and the phase's barriers
            if reads >= 84 and sum(1 for x in o if x == "s_barrier") >= 2 and (best is None or b - a < best[1] - best[0]):
                best = (a, b)
        elif 42 <= reads <= 46 and any("f64" in x for x in o):
            if best is None or b - a < best[1] - best[0]:
                best = (a, b)
    o = ops(*best)
    valu = [x for x in o if x.startswith("v_")]
    mix = Counter("half" if half_rate(x) else "plain" for x in valu)
    kinds = Counter()
    for x in valu:
        kinds["dpp" if "_dpp" in x else "add3" if "add3" in x else "f64_or_convert" if ("f64" in x or x.startswith("v_cvt")) else
              "other_half_rate" if half_rate(x) else "plain"] += 1
    ns_row = mix["plain"] * NS_PLAIN + mix["half"] * NS_HALF
    tiles = ((NX + TILE_W - 1) // TILE_W) * (NY // TH + (1 if NY % TH else 0))
    wave_rows = tiles * TH
    total_valu = None
    if len(sys.argv) > 1:
        total_valu = float(sys.argv[1])
    else:
        try:
            block = open(os.path.join(REPO, "profiles", "r06_std67_pmc_summary.txt" if std else "r06_tpi67_pmc_summary.txt")).read() \
                .split("std_ring_kernel<67, false" if std else "tpi_march_kernel<67")[1]
            total_valu = float(re.search(r"SQ_INSTS_VALU\s+n=\s*\d+\s+mean=([0-9.e+]+)", block).group(1))
        except (OSError, IndexError, AttributeError):
            pass
    row_instr = len(valu) * wave_rows
    ms_rows = ns_row * wave_rows / 1024 / 1e6
    result = {
        "kernel": "std_ring_kernel<67, false, kStdMain>, phase loop (one output row per wave and phase)" if std else
                  "tpi_march_kernel<67, 60, 12, true, true, true>, row loop (whole-metre tiles)",
        "row_loop_instructions": {"valu": len(valu), "by_kind": dict(kinds), "ds_read_b128": sum(1 for x in o if x == "ds_read_b128"),
                                  "salu": sum(1 for x in o if x.startswith("s_"))},
        "issue_cost_ns_per_wave_instruction_and_simd": {"plain": NS_PLAIN, "half_rate": NS_HALF,
                                                        "source": "profiles/r02_valu_mix_rate.txt, profiles/r03_valu_mix2_rate.txt (16 waves per CU, wall clock)"},
        "ns_per_wave_row": round(ns_row, 1),
        "wave_rows_per_launch": wave_rows,
        "row_loop_valu_bound_ms": round(ms_rows, 3),
    }
    if total_valu and std:
        # the phase loop holds branches only some waves take (waves 0-3 stage, convert and write the ring rows for all
        # twelve), so its static count overstates a wave's work: the launch's counter, priced at the loop's mix, is the bound
        avg = ns_row / len(valu)
        result.update({"SQ_INSTS_VALU_per_launch": total_valu,
                       "static_count_x_wave_rows": row_instr,
                       "note": "the static count includes the staging share only waves 0-3 execute; the bound prices the "
                               "launch's SQ_INSTS_VALU at the phase loop's mix of issue classes",
                       "ns_per_valu_instruction_at_this_mix": round(avg, 3),
                       "valu_bound_ms": round(total_valu * avg / 1024 / 1e6, 3)})
    elif total_valu:
        rest = max(0.0, total_valu - row_instr)
        avg = ns_row / len(valu)
        ms_rest = rest * avg / 1024 / 1e6
        result.update({"SQ_INSTS_VALU_per_launch": total_valu, "valu_outside_the_row_loop": rest,
                       "outside_valu_bound_ms": round(ms_rest, 3), "valu_bound_ms": round(ms_rows + ms_rest, 3)})
    else:
        result["valu_bound_ms"] = round(ms_rows, 3)
    import bench
    result["kernel_sources_sha256"] = bench.kernel_sources_sha256()
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
