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

ROUTES = ["direct", "matrix", "folded", "fft"]
ROUTE_CODE = {"direct": 0, "matrix": 1 + 4, "folded": 1 + 4 + 8, "fft": 2}


def _set_route(monkeypatch, route):
    """The evaluations of the angle loop: tap by tap in float32 (csrc/valley.hip); the dense product on the matrix pipe over the
    cells that hold a tap (csrc/valley_mfma.hip), and its folded form over PAIRS of cells, which point-symmetric tables - the
    reference's - take by default up to 17 px; the FFT that large kernels take.  The switches are read at every launch."""
    monkeypatch.setenv("TOPO_AMD_VALLEY_FFT_MIN_KERNEL", "1" if route == "fft" else "100000")
    if route == "direct":
        monkeypatch.setenv("TOPO_AMD_VALLEY_MFMA_MAX_KERNEL", "0")
    else:
        monkeypatch.delenv("TOPO_AMD_VALLEY_MFMA_MAX_KERNEL", raising=False)
    if route == "matrix":
        monkeypatch.setenv("TOPO_AMD_VALLEY_FOLD", "0")
    else:
        monkeypatch.delenv("TOPO_AMD_VALLEY_FOLD", raising=False)


VR_TAGS = ["int_valley_s7", "int_ridge_s7", "int_valley_s5", "int_valley_s17", "int_valley_s9_flat0",
           "int_ridge_s9_flat2", "frac_valley_s7", "frac_valley_s9_sig"]


def _case(g, tag):
    p = g[f"{tag}_params"]
    size, mode, sigma, flats = int(p[0]), ("valley", "ridge")[int(p[1])], (None if p[2] < 0 else float(p[2])), list(p[3:])
    dem = g["dem_int"] if tag.startswith("int") else g["dem_frac"]
    return dem, size, mode, flats, sigma


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("tag", VR_TAGS)
def test_valley_ridge_against_the_reference(golden, tag, route, monkeypatch):
    _set_route(monkeypatch, route)
    g = golden("valley_ridge")
    dem, size, mode, flats, sigma = _case(g, tag)
    norm_ref, dir_ref = g[f"{tag}_norm"], g[f"{tag}_dir"]
    norm, direction = topo.valley_ridge(dem, size, mode, flats, sigma)
    # the 17 px kernels (24 x 24 canvas, 399 cells with taps) are beyond the unfolded matrix-pipe kernel's 240 cells: tap by tap;
    # folded they are 13 K steps of pairs
    assert d.valley_route() == (0 if route == "matrix" and size == 17 else ROUTE_CODE[route])
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


@pytest.mark.parametrize("route", ["direct", "matrix", "folded"])
def test_valley_ridge_row_blocks_are_bit_identical(route, monkeypatch):
    """Row blocks with ghost rows (the reach of the largest rotated kernel) against the single
    block; the standardisation uses the mean / std of the whole DEM in both."""
    _set_route(monkeypatch, route)
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
    assert d.valley_route() == ROUTE_CODE[route]
    for nb in (2, 3, 7):
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


@pytest.mark.parametrize("route", ROUTES)
def test_single_rank_shard_valley_ridge(route, monkeypatch):
    """topo_amd_shard_valley_ridge with one rank: moments and standardisation on the device, the
    exchange a no-op, interior / seam split still run.  Against the float64 oracle, and equal to
    the device-block call given the same mean / std (bit for bit on the direct kernel; to rounding by
    FFT, where the interior and the seam strips are transformed separately)."""
    _set_route(monkeypatch, route)
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
    assert d.valley_route() == ROUTE_CODE[route]
    if route != "fft":
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
    monkeypatch.setenv("TOPO_AMD_VALLEY_MFMA_MAX_KERNEL", "0")
    monkeypatch.setenv("TOPO_AMD_VALLEY_FFT_MIN_KERNEL", "100000")
    norm_d, dir_d = _block_run(dem, taps, ksize, ang, 3, 1)
    monkeypatch.setenv("TOPO_AMD_VALLEY_FFT_MIN_KERNEL", "1")
    norm_f, dir_f = _block_run(dem, taps, ksize, ang, 3, 1)
    scale = float(norm_d.max())
    assert np.max(np.abs(norm_f - norm_d)) <= 2e-5 * scale
    assert np.mean(dir_f == dir_d) >= 0.995


@pytest.mark.parametrize("form", ["matrix", "folded"])
@pytest.mark.parametrize("size,planes", [(3, 3), (5, 3), (7, 1), (7, 2), (7, 3), (7, 4), (9, 3), (11, 3), (13, 3), (15, 3), (17, 3)])
def test_matrix_pipe_against_the_tap_by_tap_kernel_and_float64(size, planes, form, monkeypatch):
    """Every kernel size the matrix-pipe route takes, 1 to 4 planes, a partial last filter tile (177 angles): the norm against
    the float64 oracle and against the float32 chain, the direction through the oracle's per-angle maps.  The split-f16 product
    is the closer of the two to float64."""
    flats = [0, 0.1, 0.2, 0.3][:planes]
    dem = (orc.synthetic_dem(96, 130, seed=size) + np.random.default_rng(size).uniform(0, 1, (96, 130))).astype(np.float32)
    angles = np.arange(0, 177, 5 if size > 11 else 3 if size > 7 else 1, dtype=np.float32)
    taps, ksize, ang = topo._valley_ridge_tables(topo._valley_kernels(size, flats), angles)
    _set_route(monkeypatch, "direct")
    norm_d, dir_d = _block_run(dem, taps, ksize, ang, planes, 1)
    assert d.valley_route() == 0
    _set_route(monkeypatch, form)
    norm_m, dir_m = _block_run(dem, taps, ksize, ang, planes, 1)
    if form == "matrix" and size > 13:
        assert d.valley_route() == 0   # more than 240 cells with taps: the unfolded form hands the whole call to the tap-by-tap kernel
        return
    assert d.valley_route() == ROUTE_CODE[form]
    (norm_ex, _), maps = orc.valley_ridge_exact(dem, size, "valley", flats, angles=angles, return_maps=True)
    scale = float(np.max(norm_ex))
    err_m, err_d = float(np.max(np.abs(norm_m - norm_ex))), float(np.max(np.abs(norm_d - norm_ex)))
    assert err_m <= 2e-6 * scale and err_m <= 1.5 * err_d + 1e-7 * scale, (err_m, err_d, scale)
    index = np.searchsorted(angles, dir_m)
    assert np.all(angles[index] == dir_m)
    assert np.max(np.max(maps, axis=0) - np.take_along_axis(maps, index[None], axis=0)[0]) <= 1e-5 * scale
    assert np.mean(dir_m == dir_d) >= 0.99


@pytest.mark.parametrize("form", ["matrix", "folded"])
def test_matrix_pipe_hands_non_finite_windows_to_the_tap_by_tap_kernel(form, monkeypatch):
    """NaN, +-inf and a sample beyond the f16 range after standardising (3e9 m): exactly the pixels whose kernel footprint holds
    one of them are evaluated tap by tap (their bits are the direct route's), every other pixel by the matrix pipe (its bits are
    those of the same DEM without the specials), nothing is left marked, and row blocks keep the single block's bits."""
    size, flats = 7, [0, 0.15, 0.3]
    clean = (orc.synthetic_dem(150, 210, seed=5) + np.random.default_rng(1).uniform(0, 1, (150, 210))).astype(np.float32)
    dem = clean.copy()
    dem[40, 50] = np.nan
    dem[100:103, 150] = np.inf
    dem[0, 0] = np.nan
    dem[149, 209] = -np.inf
    dem[70, 100] = 3e9
    dem[31, 64] = np.nan            # on a tile seam of the matrix-pipe kernel (32 rows x 64 columns)
    special = ~np.isfinite(dem) | (dem > 1e9)
    angles = np.arange(0, 180, dtype=np.float32)
    taps, ksize, ang = topo._valley_ridge_tables(topo._valley_kernels(size, flats), angles)
    mean, stdev = float(clean.mean()), float(clean.std())

    def run(field, nblocks=1):
        gny, nx = field.shape
        up, down = shard.halo_rows(_lib.DESC_VALLEY_RIDGE, int(ksize.max()))
        norms, dirs = [], []
        for row0, rows in shard.split_rows(gny, nblocks):
            lo, hi = max(0, row0 - up), min(gny, row0 + rows + down)
            dev = d.DeviceArray.from_host(field[lo:hi])
            n, a = d.DeviceArray(rows, nx), d.DeviceArray(rows, nx)
            d.Block(dev, row0=lo, gny=gny).valley_ridge(taps, ksize, ang, 3, mean, stdev, n, a, out_row0=row0, out_rows=rows)
            d.sync()
            norms.append(n.to_host())
            dirs.append(a.to_host())
            for x in (dev, n, a):
                x.free()
        return np.concatenate(norms), np.concatenate(dirs)

    _set_route(monkeypatch, "direct")
    norm_d, dir_d = run(dem)
    _set_route(monkeypatch, form)
    norm_m, dir_m = run(dem)
    assert d.valley_route() == ROUTE_CODE[form]
    norm_c, dir_c = run(clean)
    assert not np.any(norm_m == -1.0)
    # the footprint: the cells of the 10 x 10 window in which some kernel has a tap, around every special sample
    kmax = int(ksize.max())
    live = np.zeros((kmax, kmax), bool)
    pos = 0
    for ks in ksize:
        t = taps[pos:pos + ks * ks * 4].reshape(ks, ks, 4)[:, :, :3]
        sh = kmax // 2 - ks // 2
        live[sh:sh + ks, sh:sh + ks] |= np.any(t != 0, axis=2)
        pos += ks * ks * 4
    touched = np.zeros(dem.shape, bool)
    for y, x in zip(*np.nonzero(special)):
        for ky, kx in zip(*np.nonzero(live)):
            oy, ox = y - (ky - kmax // 2), x - (kx - kmax // 2)   # the pixel whose window cell (ky, kx) is (y, x)
            if 0 <= oy < dem.shape[0] and 0 <= ox < dem.shape[1]:
                touched[oy, ox] = True
    assert touched.sum() > 200
    same = lambda a, b: np.array_equal(a, b, equal_nan=True)  # noqa: E731
    assert same(norm_m[touched], norm_d[touched]) and same(dir_m[touched], dir_d[touched])
    assert same(norm_m[~touched], norm_c[~touched]) and same(dir_m[~touched], dir_c[~touched])
    for nb in (2, 5):
        norm_b, dir_b = run(dem, nb)
        assert same(norm_b, norm_m) and same(dir_b, dir_m), nb


def test_tables_that_are_not_point_symmetric_take_the_unfolded_form(monkeypatch):
    """The folded form is for tables that are point-symmetric bit by bit (the reference's are); one tap changed and the call runs
    over the cells instead of the pairs - with that tap honoured."""
    dem = (orc.synthetic_dem(80, 100, seed=2) + np.random.default_rng(2).uniform(0, 1, (80, 100))).astype(np.float32)
    angles = np.arange(0, 180, 4, dtype=np.float32)
    taps, ksize, ang = topo._valley_ridge_tables(topo._valley_kernels(7, [0, 0.15, 0.3]), angles)
    _set_route(monkeypatch, "folded")
    norm_f, _ = _block_run(dem, taps, ksize, ang, 3, 1)
    assert d.valley_route() == 13
    bent = taps.copy()
    first = np.nonzero(bent)[0][0]
    bent[first] *= 1.5
    norm_b, dir_b = _block_run(dem, bent, ksize, ang, 3, 1)
    assert d.valley_route() == 5
    _set_route(monkeypatch, "direct")
    norm_d, dir_d = _block_run(dem, bent, ksize, ang, 3, 1)
    scale = float(norm_d.max())
    assert np.max(np.abs(norm_b - norm_d)) <= 1e-5 * scale and np.mean(dir_b == dir_d) >= 0.99
    assert np.max(np.abs(norm_b - norm_f)) > 1e-4 * scale   # and the changed tap shows


@pytest.mark.parametrize("size,planes", [(19, 3), (21, 1), (25, 3), (33, 2), (41, 3), (43, 4), (47, 3), (65, 3), (83, 2)])
def test_streamed_matrix_pipe_form_for_kernels_of_19_to_85_px(size, planes, monkeypatch):
    """Kernels whose pairs of cells no longer fit a wave's registers (more than 15 K steps) stream their pixel operands chunk by
    chunk (valley_fold_stream_kernel): against the float64 oracle and the tap-by-tap kernel, row blocks bit-identical, non-finite
    samples handed over pixel by pixel."""
    flats = [0, 0.1, 0.2, 0.3][:planes]
    dem = (orc.synthetic_dem(120, 150, seed=size) + np.random.default_rng(size).uniform(0, 1, (120, 150))).astype(np.float32)
    angles = np.arange(0, 178, 7 if size < 47 else 19, dtype=np.float32)
    taps, ksize, ang = topo._valley_ridge_tables(topo._valley_kernels(size, flats), angles)
    assert 25 < ksize.max() <= 120
    _set_route(monkeypatch, "direct")
    norm_d, dir_d = _block_run(dem, taps, ksize, ang, planes, 1)
    assert d.valley_route() == 0
    _set_route(monkeypatch, "folded")
    norm_m, dir_m = _block_run(dem, taps, ksize, ang, planes, 1)
    assert d.valley_route() == 1 + 4 + 8 + 16
    (norm_ex, _), maps = orc.valley_ridge_exact(dem, size, "valley", flats, angles=angles, return_maps=True)
    scale = float(np.max(norm_ex))
    err_m, err_d = float(np.max(np.abs(norm_m - norm_ex))), float(np.max(np.abs(norm_d - norm_ex)))
    assert err_m <= 3e-6 * scale and err_m <= 1.5 * err_d + 1e-7 * scale, (err_m, err_d, scale)
    index = np.searchsorted(angles, dir_m)
    assert np.all(angles[index] == dir_m)
    assert np.max(np.max(maps, axis=0) - np.take_along_axis(maps, index[None], axis=0)[0]) <= 1e-5 * scale
    assert np.mean(dir_m == dir_d) >= 0.99
    for nb in (2, 3):
        norm_b, dir_b = _block_run(dem, taps, ksize, ang, planes, nb)
        assert np.array_equal(norm_b, norm_m) and np.array_equal(dir_b, dir_m), nb
    # a NaN and an infinity: the pixels they reach take the tap-by-tap kernel's bits, nothing stays marked
    holed = dem.copy()
    holed[60, 70] = np.nan
    holed[5, 140] = np.inf
    mean, stdev = float(dem.mean()), float(dem.std())

    def run(route):
        _set_route(monkeypatch, route)
        dev = d.DeviceArray.from_host(holed)
        n, a = d.DeviceArray(*holed.shape), d.DeviceArray(*holed.shape)
        d.Block(dev).valley_ridge(taps, ksize, ang, planes, mean, stdev, n, a)
        d.sync()
        out = n.to_host(), a.to_host()
        for x in (dev, n, a):
            x.free()
        return out

    nh_d, dh_d = run("direct")
    nh_m, dh_m = run("folded")
    assert not np.any(nh_m == -1.0)
    reach = int(ksize.max()) // 2 + 1
    near = np.zeros(dem.shape, bool)
    near[max(0, 60 - reach):60 + reach + 1, max(0, 70 - reach):70 + reach + 1] = True
    near[0:5 + reach + 1, max(0, 140 - reach):] = True
    differs = ~((nh_m == nh_d) | (np.isnan(nh_m) & np.isnan(nh_d)))
    touched = ~((nh_m == norm_m) | (np.isnan(nh_m) & np.isnan(norm_m)))      # pixels the two samples changed at all
    assert not np.any(differs & touched)                                     # ... carry the tap-by-tap kernel's bits
    assert not np.any(touched & ~near)


def test_default_routes(monkeypatch):
    """Nothing set: point-symmetric tables run on the matrix pipe up to rotated kernels of 120 cells a side (registers up to 17 px,
    streamed beyond), by FFT above; TOPO_AMD_VALLEY_FFT_MIN_KERNEL, when set, takes its sizes away from the matrix pipe."""
    for name in ("TOPO_AMD_VALLEY_FFT_MIN_KERNEL", "TOPO_AMD_VALLEY_MFMA_MAX_KERNEL", "TOPO_AMD_VALLEY_FOLD"):
        monkeypatch.delenv(name, raising=False)
    dem = orc.synthetic_dem(140, 160, seed=3)
    angles = np.array([0, 30, 45, 100], dtype=np.float32)
    for size, want in ((7, 13), (21, 29), (65, 29), (151, 2)):
        taps, ksize, ang = topo._valley_ridge_tables(topo._valley_kernels(size, [0, 0.15, 0.3]), angles)
        _block_run(dem, taps, ksize, ang, 3, 1)
        assert d.valley_route() == want, (size, int(ksize.max()))
    monkeypatch.setenv("TOPO_AMD_VALLEY_FFT_MIN_KERNEL", "64")
    taps, ksize, ang = topo._valley_ridge_tables(topo._valley_kernels(65, [0, 0.15, 0.3]), angles)
    _block_run(dem, taps, ksize, ang, 3, 1)
    assert d.valley_route() == 2
