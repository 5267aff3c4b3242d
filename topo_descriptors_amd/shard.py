"""Row-block sharding of a DEM over the GPUs of one node (host-side plan + device driver).

The reference's only precedent is ``dask.array.map_overlap(conv_fn, dem, depth=2*size,
boundary="none")`` for TPI (reference topo.py:177-178): independent blocks plus ghost rows, no
boundary rule at interior seams.  Here rank r owns a contiguous block of rows; before a
descriptor runs it receives ``halo_above`` ghost rows from rank r-1 and ``halo_below`` from
rank r+1 (RCCL send/recv over xGMI inside libtopo_amd, ``topo_amd_halo_exchange_start``), and the
kernels apply the descriptor's boundary rule at the GLOBAL top/bottom only, so the sharded
result is bit-identical to the single-GPU one.

``RowShardPlan`` is pure Python (tested on CPU with gloo carrying the ghost rows);
``ShardedDEM`` drives the device entry points.
"""
from dataclasses import dataclass

import numpy as np


def split_rows(gny, nranks):
    """Contiguous, balanced row ranges [(row0, rows), ...]; the first gny % nranks ranks get
    one extra row."""
    if nranks < 1 or gny < nranks:
        raise ValueError(f"cannot split {gny} rows over {nranks} ranks")
    base, extra = divmod(gny, nranks)
    out, row0 = [], 0
    for r in range(nranks):
        rows = base + (1 if r < extra else 0)
        out.append((row0, rows))
        row0 += rows
    return out


@dataclass(frozen=True)
class RowShardPlan:
    gny: int
    nx: int
    nranks: int
    rank: int
    halo_above: int   # ghost rows every rank needs from the rank above (towards row 0)
    halo_below: int

    @property
    def row0(self):
        return split_rows(self.gny, self.nranks)[self.rank][0]

    @property
    def rows_local(self):
        return split_rows(self.gny, self.nranks)[self.rank][1]

    @property
    def has_up(self):
        return self.rank > 0

    @property
    def has_down(self):
        return self.rank < self.nranks - 1

    @property
    def buffer_rows(self):
        """Rows of the device block: [halo_above | rows_local | halo_below], always allocated
        in full (the ghost rows at the global edge simply stay unused)."""
        return self.halo_above + self.rows_local + self.halo_below

    def validate(self):
        for r, (_, rows) in enumerate(split_rows(self.gny, self.nranks)):
            if rows < max(self.halo_above, self.halo_below):
                raise ValueError(
                    f"rank {r} owns {rows} rows but ghost depth is {self.halo_above}/"
                    f"{self.halo_below}: a ghost zone may not span more than one neighbour")

    # what travels: (peer, first local row, rows) to send; (peer, first buffer row, rows) to recv
    def sends(self):
        out = []
        if self.has_up and self.halo_below > 0:      # my top rows are the upper rank's bottom ghosts
            out.append((self.rank - 1, 0, self.halo_below))
        if self.has_down and self.halo_above > 0:    # my bottom rows are the lower rank's top ghosts
            out.append((self.rank + 1, self.rows_local - self.halo_above, self.halo_above))
        return out

    def recvs(self):
        out = []
        if self.has_up and self.halo_above > 0:
            out.append((self.rank - 1, 0, self.halo_above))
        if self.has_down and self.halo_below > 0:
            out.append((self.rank + 1, self.halo_above + self.rows_local, self.halo_below))
        return out

    def valid_window(self):
        """(first buffer row, rows, global row of the first) of the rows that hold real data
        after the exchange."""
        first = 0 if self.has_up else self.halo_above
        rows = self.rows_local + (self.halo_above if self.has_up else 0) + \
            (self.halo_below if self.has_down else 0)
        return first, rows, self.row0 - (self.halo_above if self.has_up else 0)

    def interior(self):
        """Owned rows [a, b) (global) whose stencil never touches a ghost row: these are
        computed while the exchange is in flight."""
        a = self.row0 + (self.halo_above if self.has_up else 0)
        b = self.row0 + self.rows_local - (self.halo_below if self.has_down else 0)
        a = min(a, self.row0 + self.rows_local)
        return a, max(a, b)


def halo_rows(descriptor, p0=0.0, p1=0.0):
    """(above, below) ghost depth of a descriptor, from the library (topo_amd_halo_rows)."""
    import ctypes as C

    from . import _lib
    up, down = C.c_int32(), C.c_int32()
    _lib.check(_lib.load().topo_amd_halo_rows(int(descriptor), float(p0), float(p1), C.byref(up),
                                              C.byref(down)), "topo_amd_halo_rows")
    return up.value, down.value


class ShardedDEM:
    """One rank's block of a row-sharded DEM on its GPU, with ghost rows refreshed over RCCL.

    The communicator must exist (``init_comm``).  Every descriptor call is collective: all
    ranks call it with the same parameters."""

    def __init__(self, plan, local_rows=None):
        from . import device as d
        from . import _lib
        self.plan = plan
        plan.validate()
        self.block = d.DeviceArray(plan.buffer_rows, plan.nx)
        # the ghost rows are read by nobody before an exchange has filled them, but a stale NaN in recycled
        # memory should not be what a debugging session finds there
        _lib.check(_lib.lib().topo_amd_memset(self.block.ptr, 0, self.block.nbytes), "topo_amd_memset")
        if local_rows is not None:
            self.block.upload_rows(local_rows, plan.halo_above)

    def classify(self):
        """Collective: the class of the WHOLE raster from the lattice samples of all shards
        (``topo_amd_shard_classify``), so that every shard takes the kernels the single-GPU run takes.  The descriptor
        calls do this themselves at the first call after the shard's rows were written through the library; calling it
        is only needed when the rows were rewritten behind the library's back."""
        from . import _lib
        p = self.plan
        _lib.check(_lib.lib().topo_amd_shard_classify(self.block.row_ptr(p.halo_above), p.rows_local, p.row0, p.gny, p.nx),
                   "topo_amd_shard_classify")

    def _collective(self, name, *args):
        """Call a ``topo_amd_shard_*`` entry point with this buffer's ghost depth declared
        (``topo_amd_shard_layout``): a descriptor that needs fewer ghost rows than the plan reserves uses the
        ones next to the owned rows, one that needs more is refused instead of reading the owned rows from the
        wrong offset and receiving past the end of the buffer."""
        import ctypes as C

        from . import _lib
        lib = _lib.lib()
        # the declaration is per thread in the library; whatever this thread had declared before (the application
        # may drive shards of its own through the C ABI) is put back afterwards
        was_up, was_down = C.c_int32(), C.c_int32()
        _lib.check(lib.topo_amd_shard_layout_get(C.byref(was_up), C.byref(was_down)), "shard_layout_get")
        _lib.check(lib.topo_amd_shard_layout(self.plan.halo_above, self.plan.halo_below), "shard_layout")
        try:
            _lib.check(getattr(lib, name)(*args), name)
        finally:
            lib.topo_amd_shard_layout(was_up.value, was_down.value)

    @staticmethod
    def init_comm(rank, nranks, broadcast_bytes):
        """``broadcast_bytes(payload_or_None) -> payload``: any out-of-band broadcast from rank 0
        (bench.py uses torch.distributed/gloo; mpi4py or a file would do as well)."""
        import ctypes as C

        from . import _lib
        lib = _lib.lib()
        uid = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        if rank == 0:
            _lib.check(lib.topo_amd_comm_unique_id(uid), "comm_unique_id")
        payload = broadcast_bytes(uid.raw if rank == 0 else None)
        _lib.check(lib.topo_amd_comm_init(rank, nranks, payload), "comm_init")

    def tpi_std(self, size, tpi=None, std=None):
        from . import _lib
        p = self.plan
        self._collective("topo_amd_shard_tpi_std", self.block.ptr, p.rows_local, p.row0, p.gny, p.nx,
                                                     int(size), tpi.ptr if tpi else None,
                                                     std.ptr if std else None)

    def gradient(self, sigma, res_x, res_y, sig_ratio=1.0, dx=None, dy=None, slope=None, aspect=None):
        from . import _lib
        p = self.plan
        rx = np.ascontiguousarray(res_x, dtype=np.float64)
        ry = np.ascontiguousarray(res_y, dtype=np.float64)
        mode = _lib.RES_SCALAR if rx.size == 1 and ry.size == 1 else _lib.RES_1D
        if mode == _lib.RES_1D and (rx.size != p.nx or ry.size != p.gny):
            raise ValueError(f"1-D resolutions must have nx = {p.nx} and gny = {p.gny} entries (the rows of the "
                             f"WHOLE DEM), got {rx.size} and {ry.size}")
        outs = [a.ptr if a else None for a in (dx, dy, slope, aspect)]
        self._collective("topo_amd_shard_gradient", self.block.ptr, p.rows_local, p.row0, p.gny, p.nx,
                                                      float(sigma), float(sig_ratio), mode,
                                                      _lib.ptr(rx), _lib.ptr(ry), *outs)

    def valley_ridge(self, taps, ksize, angles, n_planes, norm, direction):
        """Collective.  The plan's halo must be ``halo_rows(DESC_VALLEY_RIDGE, ksize.max())``; the mean
        and standard deviation of the whole DEM are formed inside (one all-reduce)."""
        from . import _lib
        p = self.plan
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        ksize = np.ascontiguousarray(ksize, dtype=np.int32)
        angles = np.ascontiguousarray(angles, dtype=np.float32)
        self._collective("topo_amd_shard_valley_ridge", self.block.ptr, p.rows_local, p.row0, p.gny, p.nx, taps.ctypes.data_as(_lib._vp),
            ksize.ctypes.data_as(_lib._i32p), angles.ctypes.data_as(_lib._vp), ksize.size, int(n_planes),
            norm.ptr, direction.ptr)

    def sx(self, dj, di, dist, window, height, out):
        from . import _lib
        p = self.plan
        dj = np.ascontiguousarray(dj, dtype=np.int32)
        di = np.ascontiguousarray(di, dtype=np.int32)
        dist = np.ascontiguousarray(dist, dtype=np.float64)
        self._collective("topo_amd_shard_sx", self.block.ptr, p.rows_local, p.row0, p.gny, p.nx,
                                                dj.ctypes.data_as(_lib._i32p),
                                                di.ctypes.data_as(_lib._i32p),
                                                dist.ctypes.data_as(_lib._f64p), dist.size, int(window),
                                                float(height), out.ptr)

    def sx_multi(self, sectors, height, outs):
        """Sx of several azimuth sectors with ONE ghost-row exchange.  The plan's halo must be the
        largest (-dj, dj) over the usable ray pixels of all sectors (``sx_multi_halo``)."""
        import ctypes as C

        from . import _lib
        from .device import pack_sectors
        p = self.plan
        first, dj, di, dist, window = pack_sectors(sectors)
        planes = (C.c_void_p * len(outs))(*[o.ptr for o in outs])
        self._collective("topo_amd_shard_sx_multi", self.block.ptr, p.rows_local, p.row0, p.gny, p.nx, len(sectors), first.ctypes.data_as(_lib._i32p),
            dj.ctypes.data_as(_lib._i32p), di.ctypes.data_as(_lib._i32p), dist.ctypes.data_as(_lib._f64p),
            window.ctypes.data_as(_lib._i32p), float(height), planes)


def sx_multi_halo(sectors):
    """(above, below) ghost depth that serves every sector of ``sectors``."""
    up = down = 0
    for _, dj, _, dist in sectors:
        dj = np.atleast_1d(dj)[~np.isnan(np.atleast_1d(dist))]
        if dj.size:
            up, down = max(up, int(-dj.min())), max(down, int(dj.max()))
    return max(up, 0), max(down, 0)
