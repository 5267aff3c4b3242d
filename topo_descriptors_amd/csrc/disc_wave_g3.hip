// Wave-shift disc kernels, instantiation group 3 (split over several translation units so the
// per-size specialisations compile in parallel).
#include "disc_wave_impl.hpp"

namespace topo {

int launch_disc_wave_group3(const Block& b, int size, float* tpi_out, float* std_out) {
    switch (size) {
        case 11: return launch_wave_any<11>(b, tpi_out, std_out);
        case 21: return launch_wave_any<21>(b, tpi_out, std_out);
        case 101: return launch_wave_any<101>(b, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
