// K1, two disc sizes in one pass over the DEM (the multi-scale half): the ring build of the larger
// disc evaluates the smaller one from the same ring (disc_ring_impl.hpp, tpi_ring_kernel<SIZE, 8, kRingMain, SIZE2>).
// Compiled for the pairs of the sizes the ring build serves on its own (5 ... 11 px, where TPI moves bytes rather
// than instructions: 2.0 ms per size at 32768^2, 4.2 TB/s); the callers' scale loops are reference topo.py:132-141
// and scripts/compute_topo_descriptors.py:25-38.  Tiles the pass leaves (fractional or non-finite samples) go to the
// general kernel of each size through the pass's tile map, as after a single-size launch.
#include "disc_wave_impl.hpp"

namespace topo {

namespace {

template <int SB, int SA>
int launch_pair(const Block& b, float* out_b, float* out_a) {
    static_assert(SB > SA, "larger disc first");
    using RC = RingCfg<SB, 8>;
    constexpr int map_tw = RGeo<SB, 8>::TILE_W;
    TOPO_TRY((launch_ring<SB, 8, kRingMain, SA>(b, out_b, out_a)));
    TOPO_TRY((launch_wave<SB, tile_rows(SB, 12, 60), 12, true, false>(b, out_b, nullptr, true, RC::TH, map_tw)));
    return launch_wave<SA, tile_rows(SA, 12, 60), 12, true, false>(b, out_a, nullptr, true, RC::TH, map_tw);
}

}  // namespace

bool tpi_pair_covers(int size_a, int size_b) {
    auto ok = [](int s) { return s == 5 || s == 7 || s == 9 || s == 11; };
    return ok(size_a) && ok(size_b) && size_a != size_b;
}

// TPI of two disc sizes from one staging pass.  TOPO_AMD_EUNSUP: no instantiation for the pair (or planes the
// 16-byte row accesses cannot take): the caller launches the sizes one by one.
int launch_tpi_pair(const Block& b, int size_a, float* out_a, int size_b, float* out_b) {
    if (b.nx % 4 != 0 || (reinterpret_cast<uintptr_t>(b.in) & 15) || (reinterpret_cast<uintptr_t>(out_a) & 15) ||
        (reinterpret_cast<uintptr_t>(out_b) & 15) || !out_a || !out_b)
        return TOPO_AMD_EUNSUP;
    if (!tpi_pair_covers(size_a, size_b)) return TOPO_AMD_EUNSUP;
    if (size_a > size_b) {
        std::swap(size_a, size_b);
        std::swap(out_a, out_b);
    }
    switch (size_b * 100 + size_a) {
        case 705: return launch_pair<7, 5>(b, out_b, out_a);
        case 905: return launch_pair<9, 5>(b, out_b, out_a);
        case 907: return launch_pair<9, 7>(b, out_b, out_a);
        case 1105: return launch_pair<11, 5>(b, out_b, out_a);
        case 1107: return launch_pair<11, 7>(b, out_b, out_a);
        case 1109: return launch_pair<11, 9>(b, out_b, out_a);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
