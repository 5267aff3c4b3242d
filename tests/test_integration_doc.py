"""The ctypes stub INTEGRATION.md shows to a maintainer of the reference must actually work: its
Python code blocks of section B are executed against the in-tree library and every stub function is
compared with the product's own ``topo.*`` (same library, same entry points: identical bits)."""
import os
import re

import numpy as np
import pytest

from oracle import topo_oracle as orc

pytestmark = pytest.mark.gpu

from topo_descriptors_amd import _lib, topo  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stub_namespace():
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    section = text[text.index("## B. "):text.index("## C. ")]
    blocks = re.findall(r"```python\n(.*?)```", section, flags=re.S)
    assert len(blocks) >= 2, "section B of INTEGRATION.md should hold the stub and the valley / ridge loop"
    code = "\n".join(blocks)
    assert 'C.CDLL("libtopo_amd.so")' in code
    code = code.replace('C.CDLL("libtopo_amd.so")', f"C.CDLL({_lib.LIB_PATH!r})")
    ns = {"_rotate_kernels": topo._rotate_kernels}  # in the reference this is topo.py:515, unchanged
    exec(compile(code, "INTEGRATION.md section B", "exec"), ns)
    return ns


def test_integration_stub_runs_and_matches_the_product():
    ns = stub_namespace()
    dem = orc.synthetic_dem(150, 192, seed=31, integer=False)
    assert np.array_equal(ns["tpi"](dem, 17), topo.tpi(dem, 17))
    assert np.array_equal(ns["tpi"](dem, 7, sigma=1.75), topo.tpi(dem, 7, sigma=1.75))
    s = ns["std"](dem, 7)
    assert s.dtype == np.float64 and np.array_equal(s, topo.std(dem, 7))
    res = {"x": np.full(192, 30.0), "y": np.full(150, -30.0)}
    for a, b in zip(ns["gradient"](dem, 3.25, res), topo.gradient(dem, 3.25, res)):
        assert np.array_equal(a, b)
    # Sx through the stub's replacement of the numba loop, with the product's host geometry
    distance = topo._sx_distance(300.0, 30.0, -30.0)
    az = np.linspace(-5.0, 5.0, 15)
    centre = np.floor(np.array(distance.shape) / 2)
    src = (centre + topo._sx_source_idx_delta(az, 300.0, 30.0, -30.0)).astype(int)
    blines = topo._sx_bresenhamlines(src, centre).astype(np.int64)
    x = 2600000.0 + 30.0 * np.arange(192)
    y = 1200000.0 - 30.0 * np.arange(150)
    want = orc.sx(dem, x, y, 0.0, 300.0)
    got = ns["sx_rolling"](dem, distance, blines, 10.0)
    assert np.max(np.abs(got - want)) <= 1e-4 * np.max(np.abs(want))
    # valley index through the stub's replacement of the angle loop
    small = orc.synthetic_dem(72, 88, seed=4)
    kernels = topo._valley_kernels(7, [0, 0.15, 0.3])
    norm, direction = ns["valley_ridge_loop"](small, kernels, small.mean(), small.std())
    want_n, want_d = topo.valley_ridge(small, 7, "valley")
    assert np.array_equal(norm, want_n) and np.array_equal(direction, want_d)
