// Wave-shift disc kernels, instantiation group 2 (split over several translation units so the
// per-size specialisations compile in parallel).
#include "disc_wave_impl.hpp"

namespace topo {

int launch_disc_wave_group2(const Block& b, int size, float* tpi_out, float* std_out) {
    switch (size) {
        case 5: return launch_wave_any<5>(b, tpi_out, std_out);
        case 33: return launch_wave_any<33>(b, tpi_out, std_out);
        case 81: return launch_wave_any<81>(b, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
