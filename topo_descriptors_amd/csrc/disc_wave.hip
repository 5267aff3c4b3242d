// K1/K2 fast path dispatch: size-specialised wave-shift kernels (disc_wave_impl.hpp), compiled in
// groups (disc_wave_g*.hip).
#include "common.hpp"

namespace topo {

int launch_disc_wave_group0(const Block& b, int size, float* tpi_out, float* std_out);
int launch_disc_wave_group1(const Block& b, int size, float* tpi_out, float* std_out);
int launch_disc_wave_group2(const Block& b, int size, float* tpi_out, float* std_out);
int launch_disc_wave_group3(const Block& b, int size, float* tpi_out, float* std_out);

// TPI alone (std_out == NULL), STD alone (tpi_out == NULL) or both fused.  Returns
// TOPO_AMD_EUNSUP when no specialisation covers the request (the caller falls back to the
// generic kernel): needs nx % 4 == 0 and 16-byte aligned planes for the 16-byte row accesses.
int launch_disc_wave(const Block& b, int size, float* tpi_out, float* std_out) {
    if (b.nx % 4 != 0 || (reinterpret_cast<uintptr_t>(b.in) & 15) || (reinterpret_cast<uintptr_t>(std_out) & 15) ||
        (reinterpret_cast<uintptr_t>(tpi_out) & 15))
        return TOPO_AMD_EUNSUP;
    int r = launch_disc_wave_group0(b, size, tpi_out, std_out);
    if (r == TOPO_AMD_EUNSUP) r = launch_disc_wave_group1(b, size, tpi_out, std_out);
    if (r == TOPO_AMD_EUNSUP) r = launch_disc_wave_group2(b, size, tpi_out, std_out);
    if (r == TOPO_AMD_EUNSUP) r = launch_disc_wave_group3(b, size, tpi_out, std_out);
    return r;
}

}  // namespace topo
