// Wave-shift disc kernels, instantiation group 1 of 10 (the per-size specialisations are split
// over several translation units so that they compile in parallel).
#include "disc_wave_impl.hpp"

namespace topo {

int launch_disc_wave_group1(const Block& b, int size, float* tpi_out, float* std_out) {
    switch (size) {
        case 7: return launch_wave_any<7>(b, tpi_out, std_out);
        case 27: return launch_wave_any<27>(b, tpi_out, std_out);
        case 47: return launch_wave_any<47>(b, tpi_out, std_out);
        case 67: return launch_wave_any<67>(b, tpi_out, std_out);
        case 87: return launch_wave_any<87>(b, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
