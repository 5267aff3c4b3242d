// Wave-shift disc kernels, instantiation group 2 of 10 (the per-size specialisations are split
// over several translation units so that they compile in parallel).
#include "disc_wave_impl.hpp"

namespace topo {

int launch_disc_wave_group2(const Block& b, int size, float* tpi_out, float* std_out) {
    switch (size) {
        case 9: return launch_wave_any<9>(b, tpi_out, std_out);
        case 29: return launch_wave_any<29>(b, tpi_out, std_out);
        case 49: return launch_wave_any<49>(b, tpi_out, std_out);
        case 69: return launch_wave_any<69>(b, tpi_out, std_out);
        case 89: return launch_wave_any<89>(b, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
