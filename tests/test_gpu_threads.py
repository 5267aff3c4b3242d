"""Entry points are re-entrant per device context (SURVEY 8b "Threading"; precedent in the reference: dask workers calling
the convolution from several threads, topo.py:177-178).  ctypes releases the GIL, so two Python threads in ``topo.*`` are two
threads inside libtopo_amd.so at once: the context's mutex (csrc/common.hpp, CallGuard) makes each call run as if alone.
Every result of the concurrent runs must have the bits of its serial run."""
import threading
import zlib

import numpy as np
import pytest

from oracle import topo_oracle as orc

pytestmark = pytest.mark.gpu

from topo_descriptors_amd import _lib, device as d, topo  # noqa: E402
import topo_descriptors_amd  # noqa: E402

ITER = 20


def crc(planes):
    return tuple(zlib.crc32(np.ascontiguousarray(p).view(np.uint8)) for p in planes)


def run_threads(jobs):
    """jobs: callables returning a tuple of CRCs; each runs ITER times in its own thread, all started together."""
    results = [[] for _ in jobs]
    errors = []
    gate = threading.Barrier(len(jobs))

    def work(k, job):
        try:
            gate.wait()
            for _ in range(ITER):
                results[k].append(job())
        except Exception as exc:  # noqa: BLE001
            errors.append((k, repr(exc)))

    threads = [threading.Thread(target=work, args=(k, job)) for k, job in enumerate(jobs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    return results


def test_two_threads_in_different_host_buffer_calls(monkeypatch):
    """Different descriptors on different shapes at once, one of them through the chunk pipeline: the device planes of
    the host-buffer entry points, the workspaces and the pipeline's streams and events are one set per context."""
    monkeypatch.setenv("TOPO_AMD_HOST_CHUNK_MB", "1")
    big = orc.synthetic_dem(3100, 1024, seed=31, integer=False)    # three chunks (960, 960, 1180 rows)
    small = orc.synthetic_dem(700, 516, seed=32, integer=True)
    wide = orc.synthetic_dem(400, 2048, seed=33, integer=True)
    res = {"x": np.float64(30.0), "y": np.float64(-30.0)}

    def job_tpi_std():
        t, s = topo.tpi_std(big, 31)
        return crc((t, s))

    def job_gradient():
        return crc(topo.gradient(small, 3.25, res))

    def job_std_then_gauss():
        return crc((topo.std(wide, 67), topo.dem(wide, 13.0)))

    jobs = [job_tpi_std, job_gradient, job_std_then_gauss]
    serial = [job() for job in jobs]
    assert d.host_chunks() == 1  # (this thread's last call: the small raster)
    topo.tpi(big, 7)
    assert d.host_chunks() == 3
    got = run_threads(jobs)
    for k, runs in enumerate(got):
        assert len(runs) == ITER and all(r == serial[k] for r in runs), (k, sum(r != serial[k] for r in runs))


def test_host_buffer_calls_next_to_device_calls():
    """One thread in topo.tpi(ndarray) (holds the context for the whole call), one driving a resident block through the *_dev
    entry points (workspaces, parameter tables, the raster-class memo), one uploading and freeing device arrays."""
    dem_h = orc.synthetic_dem(1500, 768, seed=41, integer=False)
    dem_d = orc.synthetic_dem(900, 1024, seed=42, integer=True)
    dev = d.DeviceArray.from_host(dem_d)
    blk = d.Block(dev)
    sector = d.sx_offsets(0.0, 500.0, 30.0, -30.0)
    lock_free_outs = [d.DeviceArray(900, 1024) for _ in range(3)]

    def job_host():
        return crc((topo.tpi(dem_h, 67), topo.std(dem_h, 9)))

    def job_dev():
        t, s, x = lock_free_outs
        blk.tpi_std(67, tpi=t, std=s)
        blk.sx(sector[1], sector[2], sector[3], sector[0], 10.0, x)
        d.sync()
        return crc((t.to_host(), s.to_host(), x.to_host()))

    def job_alloc():
        a = d.DeviceArray.from_host(dem_d[:200])
        o = d.DeviceArray(200, 1024)
        d.Block(a).gaussian(3.25, 3.25, o)
        d.sync()
        out = crc((o.to_host(),))
        a.free()
        o.free()
        return out

    jobs = [job_host, job_dev, job_alloc]
    serial = [job() for job in jobs]
    got = run_threads(jobs)
    for k, runs in enumerate(got):
        assert all(r == serial[k] for r in runs), (k, sum(r != serial[k] for r in runs))
    for a in lock_free_outs + [dev]:
        a.free()


def test_release_host_planes_is_exposed_and_safe_between_calls():
    dem = orc.synthetic_dem(600, 512, seed=5)
    a = topo.tpi(dem, 17)
    topo_descriptors_amd.release_host_planes()
    assert np.array_equal(topo.tpi(dem, 17), a)
    d.release_host_planes()
    d.release_host_planes()  # (nothing to free: fine)
    assert np.array_equal(topo.tpi(dem, 17), a)


def test_last_error_is_per_thread():
    lib = _lib.lib()
    seen = {}

    def bad(k, size):
        out = np.empty((4, 4), dtype=np.float32)
        rc = lib.topo_amd_tpi_f32(None, 4, 4, size, 0.0, out.ctypes.data_as(_lib._f32p))
        seen[k] = (rc, lib.topo_amd_last_error().decode())

    import ctypes as C
    up, down = C.c_int32(), C.c_int32()
    assert lib.topo_amd_halo_rows(99, 0.0, 0.0, C.byref(up), C.byref(down)) != 0  # this thread's last error
    t = threading.Thread(target=bad, args=("thread", 7))
    t.start()
    t.join()
    assert seen["thread"][0] != 0 and "bad DEM" in seen["thread"][1]
    assert "unknown descriptor" in lib.topo_amd_last_error().decode()
