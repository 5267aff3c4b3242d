#!/usr/bin/env python3
"""Instruction mix of the hot loops of the 67-px disc kernels, from the compiler's ISA (no GPU needed): the phase loop of
std_ring_kernel<67, false> (one output row of 256 staged columns per wave and phase: two chains, the staging share, the
finalisation) next to the row loop of tpi_march_kernel<67, 60, 12> (one chain).  VERDICT r04 item 2 asked which scalar
instructions the ring kernel's 488 per wave-row are (the marching kernel has 50): this prints them by opcode, with what
each group is for.

    python tools/isa_breakdown.py > profiles/r05_std67_isa_breakdown.txt
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {
    "std_ring_kernel<67, false, kStdMain> phase loop": ("_ZN4topo12_GLOBAL__N_115std_ring_kernelILi67ELb0ELi0EEEvNS0_8WaveArgsEiiNS0_7PartRunE:", 84),
    "tpi_march_kernel<67, 60, 12, true, true, true> row loop": ("_ZN4topo12_GLOBAL__N_116tpi_march_kernelILi67ELi60ELi12ELb1ELb1ELb1EEEvNS0_8WaveArgsEiiNS0_7PartRunE:", 44),
}
GROUPS = [
    ("ring-slot addresses: (s0 + k) mod R as min(b + d, b + d - RB), 3 scalar per prefix row x 42 rows (shared by the two chains)",
     lambda op, c: 0),
]


def main():
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "lab.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                        os.path.join(REPO, "tools", "ubench", "tpi_lab.hip"), "-o", asm], check=True, stderr=subprocess.DEVNULL)
        txt = open(asm).read()
    for title, (symbol, reads) in KERNELS.items():
        start = txt.index(symbol)
        lines = txt[start:txt.index("s_endpgm", start)].split("\n")
        labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for i, l in enumerate(lines):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))

        def ops(a, b):
            return [x.strip().split()[0] for x in lines[a:b + 1]
                    if x.strip() and not x.strip().startswith((".", ";", "//")) and not x.strip().endswith(":")]
        cands = [(a, b) for a, b in loops if sum(1 for x in ops(a, b) if x == "ds_read_b128") == reads]
        a, b = min(cands, key=lambda ab: ab[1] - ab[0])
        c = Counter(ops(a, b))
        total = sum(c.values())
        salu = {k: v for k, v in c.items() if k.startswith("s_")}
        valu = {k: v for k, v in c.items() if k.startswith("v_")}
        print(f"== {title}: {total} instructions, {sum(valu.values())} vector, {sum(salu.values())} scalar, "
              f"{c['ds_read_b128']} ds_read_b128, {sum(v for k, v in c.items() if k.startswith('ds_write'))} ds_write, "
              f"{sum(v for k, v in c.items() if k.startswith(('global_', 'scratch_', 'buffer_')))} global / scratch")
        print("   scalar by opcode:  " + ", ".join(f"{k} {v}" for k, v in sorted(salu.items(), key=lambda kv: -kv[1])))
        print("   vector by opcode:  " + ", ".join(f"{k} {v}" for k, v in sorted(valu.items(), key=lambda kv: -kv[1])[:24]))
        if "std_ring" in title:
            adr = c["s_min_u32"]
            print(f"   what the scalar ones are: {adr} ring rows addressed per wave-row (42 = 21 runs x top / bottom, shared by the u and")
            print(f"   the u^2 chain) x 3 = {3 * adr} (s_add, s_add, s_min: slot (s0 + k) mod R without a division; each is followed by ONE")
            print(f"   v_add_u32 per chain to reach the lane's address: {2 * adr} of the {c['v_add_u32_e32']} v_add_u32); {c['s_waitcnt']} s_waitcnt (84 LDS reads + the")
            print(f"   staging loads / stores); {c['s_nop']} s_nop (the wait state between a VALU write and a DPP read of the same register: the")
            print("   hop chain is 8 dependent adds deep per step); "
                  f"{c['s_or_b64'] + c['s_cselect_b64'] + c['s_and_b64'] + c.get('s_andn2_b64', 0)} mask operations (s_or_b64 / s_cselect_b64 / s_and_b64: the 'inside the")
            print("   DEM and the block view' predicates of the 12 staged rows, the classification ballots, the per-pixel stores);")
            print(f"   {sum(v for k, v in c.items() if k.startswith('s_cmp'))} s_cmp + {sum(v for k, v in c.items() if k.startswith('s_cbranch'))} branches (rows / tiles / the re-basing test), "
                  f"{c['s_mul_i32'] + c.get('s_mul_hi_u32', 0)} s_mul (row addresses of the 12 staging loads), {c['s_barrier']} s_barrier.")
            print("   None of it is in the way of the vector ALU by count (scalar instructions issue beside the other waves' vector")
            print("   ones); the 3 x 42 address scalars could only go with a ring whose size is a multiple of the batch (R = 80 = 8 x 10")
            print("   needs 160 KiB + the flag words: 400 bytes too many) - see DESIGN.md K2.")
        print()


if __name__ == "__main__":
    main()
