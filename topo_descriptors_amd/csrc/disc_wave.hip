// K1/K2 fast path dispatch: size-specialised wave-shift kernels (disc_wave_impl.hpp) for every
// odd disc size from 3 to 101, compiled in sixteen groups (disc_wave_group.hip, one source).
#include "common.hpp"

namespace topo {

#define TOPO_DECLARE_GROUP(g) int launch_disc_wave_group##g(const Block& b, int size, float* tpi_out, float* std_out);
TOPO_DECLARE_GROUP(0) TOPO_DECLARE_GROUP(1) TOPO_DECLARE_GROUP(2) TOPO_DECLARE_GROUP(3) TOPO_DECLARE_GROUP(4) TOPO_DECLARE_GROUP(5)
TOPO_DECLARE_GROUP(6) TOPO_DECLARE_GROUP(7) TOPO_DECLARE_GROUP(8) TOPO_DECLARE_GROUP(9) TOPO_DECLARE_GROUP(10) TOPO_DECLARE_GROUP(11)
TOPO_DECLARE_GROUP(12) TOPO_DECLARE_GROUP(13) TOPO_DECLARE_GROUP(14) TOPO_DECLARE_GROUP(15)
#undef TOPO_DECLARE_GROUP

// TPI alone (std_out == NULL), STD alone (tpi_out == NULL) or both fused.  Returns
// TOPO_AMD_EUNSUP when no specialisation covers the request (the caller falls back to the
// generic kernel): needs nx % 4 == 0 and 16-byte aligned planes for the 16-byte row accesses.
bool disc_wave_covers(int size) { return size >= 3 && size <= 101 && (size & 1) == 1; }

int launch_disc_wave(const Block& b, int size, float* tpi_out, float* std_out) {
    if (b.nx % 4 != 0 || (reinterpret_cast<uintptr_t>(b.in) & 15) || (reinterpret_cast<uintptr_t>(std_out) & 15) ||
        (reinterpret_cast<uintptr_t>(tpi_out) & 15))
        return TOPO_AMD_EUNSUP;
    if (!disc_wave_covers(size)) return TOPO_AMD_EUNSUP;
    typedef int (*GroupFn)(const Block&, int, float*, float*);
    static const GroupFn groups[16] = {launch_disc_wave_group0,  launch_disc_wave_group1,  launch_disc_wave_group2,  launch_disc_wave_group3,
                                       launch_disc_wave_group4,  launch_disc_wave_group5,  launch_disc_wave_group6,  launch_disc_wave_group7,
                                       launch_disc_wave_group8,  launch_disc_wave_group9,  launch_disc_wave_group10, launch_disc_wave_group11,
                                       launch_disc_wave_group12, launch_disc_wave_group13, launch_disc_wave_group14, launch_disc_wave_group15};
    return groups[((size - 3) / 2) % 16](b, size, tpi_out, std_out);
}

}  // namespace topo
