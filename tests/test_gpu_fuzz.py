"""The randomised parity sweeps of tools/ as tests (task 5d): fixed seed, 20 - 40 s each, run as child processes
(they size their own inputs and print one line per failure).  tools/fuzz_parity.py: topo.tpi / std / tpi_std against the
exact oracle over shapes, disc sizes and value classes (fractional, mixed, negative, nodata, NaN), plus row-block
bit-identity.  tools/fuzz_gradient_sx.py: gradient / Gaussian / Sx / valley index against the reference's own scipy
calls, plus row-block bit-identity of the gradient."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool, args", [("fuzz_parity.py", ["40", "4"]), ("fuzz_parity.py", ["20", "4", "big"]),
                                        ("fuzz_gradient_sx.py", ["40", "4"])])
def test_fuzz_tool(tool, args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *args], cwd=ROOT, capture_output=True, text=True,
                         timeout=900)
    tail = "\n".join(out.stdout.splitlines()[-45:])
    assert out.returncode == 0, tail + "\n" + out.stderr[-2000:]
    assert " 0 failures" in out.stdout, tail
