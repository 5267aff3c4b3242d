"""Valley / ridge index on the GPU against the golden vectors captured from the
real reference and against the float64 oracle.

The norm is compared directly.  The direction is an arg-max over 180 candidates that are often
equal to within rounding (neighbouring angles respond almost identically), so it is judged through
the oracle's per-angle maps: the response AT the direction the GPU chose must be the maximum up to
the tolerance of the norm itself."""
import numpy as np
import pytest

from oracle import topo_oracle as orc

pytestmark = pytest.mark.gpu

from topo_descriptors_amd import _lib, device as d, shard, topo  # noqa: E402

VR_TAGS = ["int_valley_s7", "int_ridge_s7", "int_valley_s5", "int_valley_s17", "int_valley_s9_flat0",
           "int_ridge_s9_flat2", "frac_valley_s7", "frac_valley_s9_sig"]


def _case(g, tag):
    p = g[f"{tag}_params"]
    size, mode, sigma, flats = int(p[0]), ("valley", "ridge")[int(p[1])], (None if p[2] < 0 else float(p[2])), list(p[3:])
    dem = g["dem_int"] if tag.startswith("int") else g["dem_frac"]
    return dem, size, mode, flats, sigma


# both evaluations of the angle loop: the direct kernel (what these sizes take by default) and the
# FFT route that large kernels take (TOPO_AMD_VALLEY_FFT_MIN_KERNEL is read at every launch)
@pytest.mark.parametrize("route", ["direct", "fft"])
@pytest.mark.parametrize("tag", VR_TAGS)
def test_valley_ridge_against_the_reference(golden, tag, route, monkeypatch):
    monkeypatch.setenv("TOPO_AMD_VALLEY_FFT_MIN_KERNEL", "1" if route == "fft" else "100000")
    g = golden("valley_ridge")
    dem, size, mode, flats, sigma = _case(g, tag)
    norm_ref, dir_ref = g[f"{tag}_norm"], g[f"{tag}_dir"]
    norm, direction = topo.valley_ridge(dem, size, mode, flats, sigma)
    assert norm.dtype == np.float32 and direction.dtype == np.float32 and norm.shape == dem.shape
    scale = float(np.max(np.abs(norm_ref)))
    floor = float(g[f"{tag}_norm_floor"])
    # tolerance contract of SURVEY.md section 8: range-normalised 1e-4, with the reference's own
    # float32-FFT floor next to it
    assert np.max(np.abs(norm - norm_ref)) <= floor + 1e-4 * scale, tag
    (norm_ex, _), maps = orc.valley_ridge_exact(dem, size, mode, flats, sigma, return_maps=True)
    assert np.max(np.abs(norm - norm_ex)) <= 1e-4 * scale, tag
    assert np.all((direction >= 0) & (direction <= 179) & (direction == np.round(direction)))
    at_gpu_dir = np.take_along_axis(maps, direction.astype(int)[None], axis=0)[0]
    assert np.max(np.max(maps, axis=0) - at_gpu_dir) <= 1e-4 * scale, tag
    # and in practice nearly every direction is the reference's
    assert np.mean(direction == dir_ref) >= 0.99, (tag, float(np.mean(direction == dir_ref)))


def test_valley_ridge_rejects_unknown_mode_like_the_reference():
    with pytest.raises(ValueError):
        topo.valley_ridge(np.zeros((16, 16), np.float32), 5, "canyon")


def test_valley_ridge_row_blocks_are_bit_identical():
    """Row blocks with ghost rows (the reach of the largest rotated kernel) against the single
    block; the standardisation uses the mean / std of the whole DEM in both."""
    dem = orc.synthetic_dem(150, 200, seed=9)
    size, flats = 9, [0, 0.15, 0.3]
    kernels = topo._valley_kernels(size, flats)
    taps, ksize, angles = topo._valley_ridge_tables(kernels, np.arange(0, 180, 7, dtype=np.float32))
    up, down = shard.halo_rows(_lib.DESC_VALLEY_RIDGE, int(ksize.max()))
    assert up == ksize.max() // 2 and down == ksize.max() - 1 - ksize.max() // 2
    mean, stdev = float(dem.mean()), float(dem.std())
    gny, nx = dem.shape

    def run(nblocks):
        norms, dirs = [], []
        for row0, rows in shard.split_rows(gny, nblocks):
            lo, hi = max(0, row0 - up), min(gny, row0 + rows + down)
            dev = d.DeviceArray.from_host(dem[lo:hi])
            blk = d.Block(dev, row0=lo, gny=gny)
            n, a = d.DeviceArray(rows, nx), d.DeviceArray(rows, nx)
            blk.valley_ridge(taps, ksize, angles, len(flats), mean, stdev, n, a, out_row0=row0, out_rows=rows)
            d.sync()
            norms.append(n.to_host())
            dirs.append(a.to_host())
            for x in (n, a, dev):
                x.free()
        return np.concatenate(norms), np.concatenate(dirs)

    whole = run(1)
    for nb in (2, 3):
        parts = run(nb)
        assert np.array_equal(parts[0], whole[0]) and np.array_equal(parts[1], whole[1]), nb
    # a subset of angles is what the oracle gets too
    (norm_ex, _), maps = orc.valley_ridge_exact(dem, size, "valley", flats, angles=angles, return_maps=True)
    assert np.max(np.abs(whole[0] - norm_ex)) <= 1e-4 * np.max(norm_ex)


def test_device_mean_std_matches_numpy():
    dem = orc.synthetic_dem(700, 900, seed=12)
    dev = d.DeviceArray.from_host(dem)
    mean, stdev = d.mean_std(dev)
    dev.free()
    assert abs(mean - float(np.mean(dem, dtype=np.float64))) <= 1e-9 * abs(mean)
    assert abs(stdev - float(np.std(dem, dtype=np.float64))) <= 1e-9 * stdev


@pytest.mark.parametrize("route", ["direct", "fft"])
def test_single_rank_shard_valley_ridge(route, monkeypatch):
    """topo_amd_shard_valley_ridge with one rank: moments and standardisation on the device, the
    exchange a no-op, interior / seam split still run.  Against the float64 oracle, and equal to
    the device-block call given the same mean / std (bit for bit on the direct kernel; to rounding by
    FFT, where the interior and the seam strips are transformed separately)."""
    monkeypatch.setenv("TOPO_AMD_VALLEY_FFT_MIN_KERNEL", "1" if route == "fft" else "100000")
    dem = orc.synthetic_dem(140, 192, seed=21)
    gny, nx = dem.shape
    size, flats = 7, [0, 0.15, 0.3]
    angles_in = np.arange(0, 180, 5, dtype=np.float32)
    taps, ksize, angles = topo._valley_ridge_tables(topo._valley_kernels(size, flats), angles_in)
    up, down = shard.halo_rows(_lib.DESC_VALLEY_RIDGE, int(ksize.max()))
    plan = shard.RowShardPlan(gny, nx, 1, 0, up, down)
    sd = shard.ShardedDEM(plan, dem)
    n, a = d.DeviceArray(gny, nx), d.DeviceArray(gny, nx)
    sd.valley_ridge(taps, ksize, angles, len(flats), n, a)
    d.sync()
    norm, direction = n.to_host(), a.to_host()
    (norm_ex, _), maps = orc.valley_ridge_exact(dem, size, "valley", flats, angles=angles_in, return_maps=True)
    scale = float(np.max(norm_ex))
    assert np.max(np.abs(norm - norm_ex)) <= 1e-4 * scale
    idx = np.searchsorted(angles_in, direction)
    assert np.all(angles_in[idx] == direction)
    assert np.max(np.max(maps, axis=0) - np.take_along_axis(maps, idx[None], axis=0)[0]) <= 1e-4 * scale
    # whole metres: the float64 moments are exact, so they are numpy's float64 mean / std
    dev = d.DeviceArray.from_host(dem)
    mean, stdev = d.mean_std(dev)
    assert mean == float(np.mean(dem, dtype=np.float64))
    n2, a2 = d.DeviceArray(gny, nx), d.DeviceArray(gny, nx)
    d.Block(dev).valley_ridge(taps, ksize, angles, len(flats), mean, stdev, n2, a2)
    d.sync()
    if route == "direct":
        assert np.array_equal(n2.to_host(), norm) and np.array_equal(a2.to_host(), direction)
    else:
        assert np.max(np.abs(n2.to_host() - norm)) <= 1e-5 * scale
        assert np.mean(a2.to_host() == direction) >= 0.995
    for x in (n, a, n2, a2, dev):
        x.free()


def _block_run(dem, taps, ksize, angles, n_planes, nblocks):
    gny, nx = dem.shape
    up, down = shard.halo_rows(_lib.DESC_VALLEY_RIDGE, int(ksize.max()))
    mean, stdev = float(dem.mean()), float(dem.std())
    norms, dirs = [], []
    for row0, rows in shard.split_rows(gny, nblocks):
        lo, hi = max(0, row0 - up), min(gny, row0 + rows + down)
        dev = d.DeviceArray.from_host(dem[lo:hi])
        n, a = d.DeviceArray(rows, nx), d.DeviceArray(rows, nx)
        d.Block(dev, row0=lo, gny=gny).valley_ridge(taps, ksize, angles, n_planes, mean, stdev, n, a,
                                                    out_row0=row0, out_rows=rows)
        d.sync()
        norms.append(n.to_host())
        dirs.append(a.to_host())
        for x in (dev, n, a):
            x.free()
    return np.concatenate(norms), np.concatenate(dirs)


def test_kernels_beyond_the_lds_tile_go_through_the_fft():
    """151 px: rotated kernels up to 214 px, more than the direct kernel can stage.  Against the
    float64 oracle on a handful of angles; row blocks agree to rounding (the FFT size differs)."""
    dem = orc.synthetic_dem(260, 300, seed=4)
    size, flats = 151, [0, 0.15, 0.3]
    angles = np.array([0, 20, 45, 90, 133], dtype=np.float32)
    taps, ksize, ang = topo._valley_ridge_tables(topo._valley_kernels(size, flats), angles)
    assert ksize.max() > 200
    norm, direction = _block_run(dem, taps, ksize, ang, len(flats), 1)
    (norm_ex, _), maps = orc.valley_ridge_exact(dem, size, "valley", flats, angles=angles, return_maps=True)
    scale = float(np.max(np.abs(norm_ex)))
    assert np.max(np.abs(norm - norm_ex)) <= 1e-4 * scale
    index = np.searchsorted(angles, direction)
    assert np.all(angles[index] == direction)
    at_gpu_dir = np.take_along_axis(maps, index[None], axis=0)[0]
    assert np.max(np.max(maps, axis=0) - at_gpu_dir) <= 1e-4 * scale
    norm2, direction2 = _block_run(dem, taps, ksize, ang, len(flats), 2)
    assert np.max(np.abs(norm2 - norm)) <= 1e-5 * scale
    assert np.mean(direction2 == direction) >= 0.999


def test_fft_route_equals_the_direct_kernel_to_rounding(monkeypatch):
    dem = orc.synthetic_dem(200, 240, seed=6)
    flats = [0, 0.15, 0.3]
    taps, ksize, ang = topo._valley_ridge_tables(topo._ridge_kernels(33, flats), np.arange(0, 180, 9, dtype=np.float32))
    monkeypatch.setenv("TOPO_AMD_VALLEY_FFT_MIN_KERNEL", "100000")
    norm_d, dir_d = _block_run(dem, taps, ksize, ang, 3, 1)
    monkeypatch.setenv("TOPO_AMD_VALLEY_FFT_MIN_KERNEL", "1")
    norm_f, dir_f = _block_run(dem, taps, ksize, ang, 3, 1)
    scale = float(norm_d.max())
    assert np.max(np.abs(norm_f - norm_d)) <= 2e-5 * scale
    assert np.mean(dir_f == dir_d) >= 0.995
