import os, sys, subprocess, numpy as np
if len(sys.argv) > 1:
    os.environ["TOPO_AMD_GRAD_CHUNK_MIN_ROWS"] = sys.argv[1]
    sys.path.insert(0, "/root/repo")
    import topo_descriptors_amd.topo as topo
    rng = np.random.default_rng(5)
    dem = (rng.random((3000, 2048), dtype=np.float32) * 900).astype(np.float32)
    out = topo.gradient(dem, 30.25, {"x": 50.0, "y": -50.0})
    np.save(sys.argv[2], np.stack(out))
else:
    for m, f in (("64", "/tmp/g_chunk.npy"), ("100000000", "/tmp/g_whole.npy")):
        subprocess.check_call([sys.executable, __file__, m, f])
    a, b = np.load("/tmp/g_chunk.npy"), np.load("/tmp/g_whole.npy")
    print("chunked == whole:", np.array_equal(a, b, equal_nan=True), a.shape)
