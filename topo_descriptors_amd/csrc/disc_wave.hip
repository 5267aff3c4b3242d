// K1/K2 fast path dispatch: size-specialised wave-shift kernels (disc_wave_impl.hpp) for every
// odd disc size from 3 to 101, compiled in ten groups (disc_wave_g*.hip).
#include "common.hpp"

namespace topo {

int launch_disc_wave_group0(const Block& b, int size, float* tpi_out, float* std_out);
int launch_disc_wave_group1(const Block& b, int size, float* tpi_out, float* std_out);
int launch_disc_wave_group2(const Block& b, int size, float* tpi_out, float* std_out);
int launch_disc_wave_group3(const Block& b, int size, float* tpi_out, float* std_out);
int launch_disc_wave_group4(const Block& b, int size, float* tpi_out, float* std_out);
int launch_disc_wave_group5(const Block& b, int size, float* tpi_out, float* std_out);
int launch_disc_wave_group6(const Block& b, int size, float* tpi_out, float* std_out);
int launch_disc_wave_group7(const Block& b, int size, float* tpi_out, float* std_out);
int launch_disc_wave_group8(const Block& b, int size, float* tpi_out, float* std_out);
int launch_disc_wave_group9(const Block& b, int size, float* tpi_out, float* std_out);

// TPI alone (std_out == NULL), STD alone (tpi_out == NULL) or both fused.  Returns
// TOPO_AMD_EUNSUP when no specialisation covers the request (the caller falls back to the
// generic kernel): needs nx % 4 == 0 and 16-byte aligned planes for the 16-byte row accesses.
bool disc_wave_covers(int size) { return size >= 3 && size <= 101 && (size & 1) == 1; }

int launch_disc_wave(const Block& b, int size, float* tpi_out, float* std_out) {
    if (b.nx % 4 != 0 || (reinterpret_cast<uintptr_t>(b.in) & 15) || (reinterpret_cast<uintptr_t>(std_out) & 15) ||
        (reinterpret_cast<uintptr_t>(tpi_out) & 15))
        return TOPO_AMD_EUNSUP;
    if (!disc_wave_covers(size)) return TOPO_AMD_EUNSUP;
    if (size == 3) return launch_disc_wave_group9(b, size, tpi_out, std_out);
    switch (((size - 5) / 2) % 10) {
        case 0: return launch_disc_wave_group0(b, size, tpi_out, std_out);
        case 1: return launch_disc_wave_group1(b, size, tpi_out, std_out);
        case 2: return launch_disc_wave_group2(b, size, tpi_out, std_out);
        case 3: return launch_disc_wave_group3(b, size, tpi_out, std_out);
        case 4: return launch_disc_wave_group4(b, size, tpi_out, std_out);
        case 5: return launch_disc_wave_group5(b, size, tpi_out, std_out);
        case 6: return launch_disc_wave_group6(b, size, tpi_out, std_out);
        case 7: return launch_disc_wave_group7(b, size, tpi_out, std_out);
        case 8: return launch_disc_wave_group8(b, size, tpi_out, std_out);
        case 9: return launch_disc_wave_group9(b, size, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
