"""Registers, spills and scratch of every kernel whose name contains <pattern>, over the 16 disc-kernel groups (hipcc
-Rpass-analysis=kernel-resource-usage; CPU only, ~3 min on 8 cores).  The large-disc kernels sit at the 168-register limit and
answer to small changes of their phase loops with spills: run this before and after touching them.
usage: python tools/spill_survey.py <pattern> [csrc dir]"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

pattern = sys.argv[1] if len(sys.argv) > 1 else "std_ring_kernel"
csrc = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "topo_descriptors_amd", "csrc")
KEYS = ("VGPRs:", "VGPRs Spill:", "SGPRs Spill:", "ScratchSize [bytes/lane]:")


def one(g):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", f"-DTOPO_GROUP={g}",
           "-DTOPO_NGROUPS=16", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(csrc, "disc_wave_group.hip"), "-o", f"/tmp/spill_g{g}.o"]
    return subprocess.run(cmd, capture_output=True, text=True).stderr


with ThreadPoolExecutor(max_workers=os.cpu_count() or 8) as pool:
    logs = list(pool.map(one, range(16)))
rows = {}
for log in logs:
    cur = None
    for line in log.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
        for key in KEYS:
            if cur and ("remark:     " + key) in line:
                rows[cur][key] = int(line.split(key)[1].split("[")[0])
names = sorted(rows, key=lambda n: (len(n), n))
demangled_all = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for name, demangled in zip(names, demangled_all):
    if pattern in demangled:
        short = demangled.replace("topo::(anonymous namespace)::", "").split("(")[0]
        v = rows[name]
        print(f"{short:50s} vgpr {v.get(KEYS[0]):4d}  vgpr spill {v.get(KEYS[1]):4d}  sgpr spill {v.get(KEYS[2]):4d}  scratch {v.get(KEYS[3]):4d}")
