// Wave-shift disc kernels, instantiation group 4 of 10 (the per-size specialisations are split
// over several translation units so that they compile in parallel).
#include "disc_wave_impl.hpp"

namespace topo {

int launch_disc_wave_group4(const Block& b, int size, float* tpi_out, float* std_out) {
    switch (size) {
        case 13: return launch_wave_any<13>(b, tpi_out, std_out);
        case 33: return launch_wave_any<33>(b, tpi_out, std_out);
        case 53: return launch_wave_any<53>(b, tpi_out, std_out);
        case 73: return launch_wave_any<73>(b, tpi_out, std_out);
        case 93: return launch_wave_any<93>(b, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
