// Wave-shift disc kernels, instantiation group 5 of 10 (the per-size specialisations are split
// over several translation units so that they compile in parallel).
#include "disc_wave_impl.hpp"

namespace topo {

int launch_disc_wave_group5(const Block& b, int size, float* tpi_out, float* std_out) {
    switch (size) {
        case 15: return launch_wave_any<15>(b, tpi_out, std_out);
        case 35: return launch_wave_any<35>(b, tpi_out, std_out);
        case 55: return launch_wave_any<55>(b, tpi_out, std_out);
        case 75: return launch_wave_any<75>(b, tpi_out, std_out);
        case 95: return launch_wave_any<95>(b, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
