import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from topo_descriptors_amd import topo
from oracle import topo_oracle as orc
sigma = 2.25
w, R = orc.gaussian_weights(sigma)
dem = np.zeros((128, 160), np.float32)
dem[40, 50] = 1024.0   # row 40: tile 32..63, middle row 48 -> c = 0 for the column
got = topo.dem(dem, (sigma, 0.0)).astype(np.float64)
col = got[40 - R:40 + R + 1, 50]
ex = 1024.0 * w[::-1]
print("impulse axis0: got/exact-1 per tap:", np.array2string((col / ex - 1), precision=2))
dem = np.zeros((128, 160), np.float32)
dem[48, 50] = 1024.0   # the impulse is the offset itself: c = 1024, d = -1024 elsewhere
got = topo.dem(dem, (sigma, 0.0)).astype(np.float64)
col = got[48 - R:48 + R + 1, 50]
print("impulse at the offset row: err:", np.array2string(col - ex, precision=3))
# a constant-slope ramp: exact result is the ramp itself away from the edges
dem = (np.arange(128, dtype=np.float32)[:, None] * 3.0 + np.zeros((1, 160), np.float32)).astype(np.float32)
got = topo.dem(dem, (sigma, 0.0)).astype(np.float64)
print("ramp err rows 20..100:", np.array2string(got[20:100, 7] - dem[20:100, 7], precision=2, max_line_width=200))
