// Wave-shift disc kernels, instantiation group 6 of 10 (the per-size specialisations are split
// over several translation units so that they compile in parallel).
#include "disc_wave_impl.hpp"

namespace topo {

int launch_disc_wave_group6(const Block& b, int size, float* tpi_out, float* std_out) {
    switch (size) {
        case 17: return launch_wave_any<17>(b, tpi_out, std_out);
        case 37: return launch_wave_any<37>(b, tpi_out, std_out);
        case 57: return launch_wave_any<57>(b, tpi_out, std_out);
        case 77: return launch_wave_any<77>(b, tpi_out, std_out);
        case 97: return launch_wave_any<97>(b, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
