// Wave-shift disc kernels, instantiation group 0 of 10 (the per-size specialisations are split
// over several translation units so that they compile in parallel).
#include "disc_wave_impl.hpp"

namespace topo {

int launch_disc_wave_group0(const Block& b, int size, float* tpi_out, float* std_out) {
    switch (size) {
        case 5: return launch_wave_any<5>(b, tpi_out, std_out);
        case 25: return launch_wave_any<25>(b, tpi_out, std_out);
        case 45: return launch_wave_any<45>(b, tpi_out, std_out);
        case 65: return launch_wave_any<65>(b, tpi_out, std_out);
        case 85: return launch_wave_any<85>(b, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
