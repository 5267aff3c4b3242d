// Wave-shift disc kernels, instantiation group 3 of 10 (the per-size specialisations are split
// over several translation units so that they compile in parallel).
#include "disc_wave_impl.hpp"

namespace topo {

int launch_disc_wave_group3(const Block& b, int size, float* tpi_out, float* std_out) {
    switch (size) {
        case 11: return launch_wave_any<11>(b, tpi_out, std_out);
        case 31: return launch_wave_any<31>(b, tpi_out, std_out);
        case 51: return launch_wave_any<51>(b, tpi_out, std_out);
        case 71: return launch_wave_any<71>(b, tpi_out, std_out);
        case 91: return launch_wave_any<91>(b, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
