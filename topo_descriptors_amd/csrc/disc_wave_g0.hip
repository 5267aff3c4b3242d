// Wave-shift disc kernels, instantiation group 0 (split over several translation units so the
// per-size specialisations compile in parallel).
#include "disc_wave_impl.hpp"

namespace topo {

int launch_disc_wave_group0(const Block& b, int size, float* tpi_out, float* std_out) {
    switch (size) {
        case 7: return launch_wave_any<7>(b, tpi_out, std_out);
        case 67: return launch_wave_any<67>(b, tpi_out, std_out);
        case 13: return launch_wave_any<13>(b, tpi_out, std_out);
        case 41: return launch_wave_any<41>(b, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
