"""Builds libtopo_amd.so in-tree with hipcc for gfx950 (no JIT cache, no pip install)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtopo_amd.so")
NGROUPS = 16  # the per-size disc kernels: one source (disc_wave_group.hip), compiled once per group of sizes
# (source, extra flags, object name); the long units first, so that the pool's last jobs are short ones
UNITS = [("sx.hip", [], "sx.o"), ("gauss.hip", [], "gauss.o")] + \
        [("disc_wave_group.hip", [f"-DTOPO_GROUP={g}", f"-DTOPO_NGROUPS={NGROUPS}"], f"disc_wave_group{g}.o") for g in range(NGROUPS)] + \
        [(s, [], s.replace(".hip", ".o")) for s in ("disc_pair.hip", "disc.hip", "disc_wave.hip", "disc_big.hip", "valley.hip",
                                                    "valley_mfma.hip", "valley_fft.hip", "capi.hip")]
SOURCES = sorted({u[0] for u in UNITS})
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (looked on PATH and in /opt/rocm/bin)")
    return exe


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


STAMP = LIB + ".srchash"  # what the library was built from: travels with it (git-ignored, not gpurun-ignored)


def sources_hash(paths):
    """sha256 over the sources, the headers and the build recipe: modification times do not survive a checkout or a
    snapshot onto another machine, the content does."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join([ARCH] + FLAGS + [str(NGROUPS)]).encode())
    for path in sorted(paths):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build_library(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the shared library.  Returns its path."""
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))
    headers.append(os.path.join(os.path.dirname(HERE), "include", "topo_amd.h"))
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    want = sources_hash(srcs + headers)
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == want:
                return LIB  # built from exactly these sources (whatever the files' times say)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()

    def deps(path, seen):
        """`path` and every quoted include it reaches (csrc/ and include/)."""
        if path in seen or not os.path.exists(path):
            return seen
        seen.add(path)
        with open(path) as fh:
            for line in fh:
                line = line.strip()
                if line.startswith("#include \""):
                    name = line.split("\"")[1]
                    for base in (os.path.dirname(path), CSRC, os.path.join(os.path.dirname(HERE), "include")):
                        deps(os.path.join(base, name), seen)
        return seen

    def compile_one(unit):
        src, extra, name = os.path.join(CSRC, unit[0]), unit[1], unit[2]
        obj = os.path.join(objdir, name)
        if not force and _newer(obj, sorted(deps(src, set())) + [__file__]):
            return obj  # object newer than its source and every header it includes
        cmd = [cc, f"--offload-arch={ARCH}", *FLAGS, *extra, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 8) as pool:
        objs = list(pool.map(compile_one, UNITS))
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs,
           "-L/opt/rocm/lib", "-lrccl", "-lhipfft", "-pthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as fh:
        fh.write(want + "\n")
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
