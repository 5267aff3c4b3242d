// Wave-shift disc kernels, instantiation group 8 of 10 (the per-size specialisations are split
// over several translation units so that they compile in parallel).
#include "disc_wave_impl.hpp"

namespace topo {

int launch_disc_wave_group8(const Block& b, int size, float* tpi_out, float* std_out) {
    switch (size) {
        case 21: return launch_wave_any<21>(b, tpi_out, std_out);
        case 41: return launch_wave_any<41>(b, tpi_out, std_out);
        case 61: return launch_wave_any<61>(b, tpi_out, std_out);
        case 81: return launch_wave_any<81>(b, tpi_out, std_out);
        case 101: return launch_wave_any<101>(b, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
