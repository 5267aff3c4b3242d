"""topo_amd_shutdown followed by a new initialisation, with FFT plans cached in between: the second
run must reproduce the first.  usage: shutdown_reinit_check.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ["TOPO_AMD_VALLEY_FFT_MIN_KERNEL"] = "1"
from topo_descriptors_amd import _lib, topo
from oracle import topo_oracle as orc
dem = orc.synthetic_dem(120, 150, seed=1)
a = topo.valley_ridge(dem, 9, "valley")
_lib.check(_lib.lib().topo_amd_shutdown(), "shutdown")
_lib._ready = False
b = topo.valley_ridge(dem, 9, "valley")
t = topo.tpi(dem, 7)
assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
print("shutdown / re-init with cached FFT plans: ok", float(a[0].max()))
