// Wave-shift disc kernels, instantiation group 9 of 10 (the per-size specialisations are split
// over several translation units so that they compile in parallel).
#include "disc_wave_impl.hpp"

namespace topo {

int launch_disc_wave_group9(const Block& b, int size, float* tpi_out, float* std_out) {
    switch (size) {
        case 3: return launch_wave_any<3>(b, tpi_out, std_out);
        case 23: return launch_wave_any<23>(b, tpi_out, std_out);
        case 43: return launch_wave_any<43>(b, tpi_out, std_out);
        case 63: return launch_wave_any<63>(b, tpi_out, std_out);
        case 83: return launch_wave_any<83>(b, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
