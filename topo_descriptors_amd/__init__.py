"""MI355X-native per-pixel topographic descriptors (tpi / std / gradient / sx).

Drop-in for the hot path of MeteoSwiss/topo-descriptors: ``topo_descriptors_amd.topo`` keeps
the reference's ``topo.*`` call signatures (reference ``topo_descriptors/topo.py``) and runs
them as hand-written HIP kernels for gfx950 through the C ABI of ``libtopo_amd.so``
(``include/topo_amd.h``).  There is no CPU fallback: without the built library and a GPU the
descriptor functions raise.
"""

__version__ = "0.1.0"


class _Config:
    """The two constants of the reference's topo_descriptors.conf (config/...conf:1-5)."""

    min_elevation = -100  # values <= min_elevation are filtered out on ingest
    scale_std = 4         # standard deviations per unit scale


CFG = _Config()

from . import helpers, topo  # noqa: E402,F401


def release_host_planes():
    """Free the device planes that ``topo.tpi(ndarray)`` and the other host-buffer calls keep between calls (grow-only; a
    32768 x 32768 gradient leaves about 20 GiB on the GPU).  ``topo_amd_release_host_planes`` of the C ABI."""
    from . import device  # noqa: PLC0415
    device.release_host_planes()
