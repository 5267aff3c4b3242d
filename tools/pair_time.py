"""Two disc sizes in one pass over the DEM (topo_amd_tpi_multi_dev, csrc/disc_pair.hip) against the two single
launches, on the bench DEM: ms per pair (HIP events, median of 10) and CRC-32 of the planes of both routes."""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topo_descriptors_amd import device as d  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dem = d.synth_dem(n, n, seed=0)
blk = d.Block(dem)
a, b = d.DeviceArray(n, n), d.DeviceArray(n, n)


def med(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


for sizes in ((5, 7), (7, 11), (9, 11), (5, 11)):
    t_pair = med(d.time_launches(lambda: blk.tpi_multi(sizes, [a, b]), 10))
    crc_pair = [zlib.crc32(x.to_host(0, 2048).tobytes()) for x in (a, b)]

    def singles():
        blk.tpi_std(sizes[0], tpi=a)
        blk.tpi_std(sizes[1], tpi=b)

    t_single = med(d.time_launches(singles, 10))
    crc_single = [zlib.crc32(x.to_host(0, 2048).tobytes()) for x in (a, b)]
    print(f"sizes {sizes}: pair kernel {t_pair:.3f} ms, two single launches {t_single:.3f} ms, "
          f"ratio {t_single / t_pair:.2f}, same bits {crc_pair == crc_single}")
