// Wave-shift disc kernels, instantiation group 1 (split over several translation units so the
// per-size specialisations compile in parallel).
#include "disc_wave_impl.hpp"

namespace topo {

int launch_disc_wave_group1(const Block& b, int size, float* tpi_out, float* std_out) {
    switch (size) {
        case 17: return launch_wave_any<17>(b, tpi_out, std_out);
        case 65: return launch_wave_any<65>(b, tpi_out, std_out);
        case 9: return launch_wave_any<9>(b, tpi_out, std_out);
        case 51: return launch_wave_any<51>(b, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
