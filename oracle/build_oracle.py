#!/usr/bin/env python3
"""Builds oracle/libtopo_oracle.so (the C/OpenMP twin of the CPU oracle) with gcc.
Run by __graft_entry__.build(); building the checker is not using it."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "topo_oracle.c")
LIB = os.path.join(HERE, "libtopo_oracle.so")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found")
    cmd = [gcc, "-O3", "-mavx2", "-mfma", "-fopenmp", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"]
    print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
