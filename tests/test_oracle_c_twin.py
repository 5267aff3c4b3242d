"""The C/OpenMP twin of the oracle against the numpy oracle (which is pinned to the reference)."""
import numpy as np
import pytest

from oracle import c_twin, topo_oracle as orc


@pytest.mark.parametrize("size", [1, 3, 6, 7, 17, 67])
def test_c_twin_tpi_std(size):
    for integer in (True, False):
        dem = orc.synthetic_dem(90, 110, seed=size, integer=integer)
        t, s = c_twin.tpi_std(dem, size)
        if size > 1:
            assert np.allclose(t, orc.tpi_exact(dem, size), rtol=0, atol=1e-9)
            assert np.allclose(s, orc.std_exact(dem, size), rtol=0, atol=1e-7)


def test_c_twin_sx(golden):
    g = golden("sx")
    dem = g["dem"]
    window, offs, dist = orc.sx_geometry(0.0, 500.0, 30.0, -30.0)
    got = c_twin.sx(dem, offs[:, 0], offs[:, 1], dist, window, 10.0)
    assert np.max(np.abs(got - g["az0_out"])) <= 2e-5
    assert c_twin.threads() >= 1
