// Wave-shift disc kernels, instantiation group 7 of 10 (the per-size specialisations are split
// over several translation units so that they compile in parallel).
#include "disc_wave_impl.hpp"

namespace topo {

int launch_disc_wave_group7(const Block& b, int size, float* tpi_out, float* std_out) {
    switch (size) {
        case 19: return launch_wave_any<19>(b, tpi_out, std_out);
        case 39: return launch_wave_any<39>(b, tpi_out, std_out);
        case 59: return launch_wave_any<59>(b, tpi_out, std_out);
        case 79: return launch_wave_any<79>(b, tpi_out, std_out);
        case 99: return launch_wave_any<99>(b, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
