// A/B for the non-output write traffic of the exact one-pass TPI kernel (67 px, integer DEM).
//
//   variant 12: disc_wave_kernel<67, 60, 12, true, false>  - the product's choice; 168-VGPR budget,
//               22 VGPRs spilled to scratch (8 dword stores per lane per tile inside the tile loop)
//   variant  8: disc_wave_kernel<67, 64,  8, true, false>  - 256-VGPR budget
//   variant  0: the product's pair - FAST 12-wave build (no fall-back paths, no scratch), then the
//               general 12-wave build over the tiles the fast one deferred
//
// Both write the same ny x nx float32 output, so under `rocprofv3 --pmc WRITE_SIZE` anything above
// ny*nx*4 bytes is not output.  Links against the product library for the context/workspace; the
// kernels are instantiated here from the same header the product uses.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench/tpi_write_ab.hip \
//         -Ltopo_descriptors_amd -ltopo_amd -Wl,-rpath,'$ORIGIN/../../topo_descriptors_amd' \
//         -o tools/ubench/tpi_write_ab
//   tools/ubench/tpi_write_ab <0|12|8> [n=32768] [reps=3] [integer=1]
#include <cstdio>
#include <cstdlib>

#include "../../include/topo_amd.h"
#include "../../topo_descriptors_amd/csrc/disc_wave_impl.hpp"

#define CK(x)                                                                         \
    do {                                                                              \
        int rc_ = (x);                                                                \
        if (rc_ != TOPO_AMD_OK) {                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #x, rc_, topo_amd_last_error());        \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 12;
    const int n = argc > 2 ? atoi(argv[2]) : 32768;
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    const int integer = argc > 4 ? atoi(argv[4]) : 1;
    CK(topo_amd_init(0));
    void *dem = nullptr, *out = nullptr;
    CK(topo_amd_malloc(&dem, (size_t)n * n * 4));
    CK(topo_amd_malloc(&out, (size_t)n * n * 4));
    CK(topo_amd_synth_dem_dev((float*)dem, n, 0, n, 0, integer));
    topo::Block b{(const float*)dem, n, 0, n, n, 0, n};
    auto run = [&]() {
        if (variant == 8) return topo::launch_wave<67, 64, 8, true, false>(b, (float*)out, nullptr);
        if (variant == 12) return topo::launch_wave<67, 60, 12, true, false>(b, (float*)out, nullptr);
        // variant 0: what the product launches - fast build, then the general build over the deferred tiles
        int rc = topo::launch_wave<67, 60, 12, true, false, true>(b, (float*)out, nullptr);
        if (rc != TOPO_AMD_OK) return rc;
        return topo::launch_wave<67, 60, 12, true, false, false>(b, (float*)out, nullptr, true);
    };
    CK(run());
    CK(topo_amd_sync());
    float ms = 0.f;
    CK(topo_amd_timer_start());
    for (int r = 0; r < reps; ++r) CK(run());
    CK(topo_amd_timer_stop(&ms));
    // checksum so the two variants can be compared for identical output
    const size_t probe = (size_t)n * 1024;
    float* h = (float*)malloc(probe * 4);
    CK(topo_amd_memcpy_d2h(h, (const char*)out + ((size_t)n / 2) * n * 4, probe * 4));
    double s = 0.0;
    for (size_t i = 0; i < probe; ++i) s += (double)h[i] * (double)((i % 97) + 1);
    printf("{\"variant\": %d, \"integer_dem\": %d, \"n\": %d, \"ms\": %.4f, \"output_bytes\": %zu, \"checksum\": %.6f}\n", variant, integer, n,
           ms / reps, (size_t)n * n * 4, s);
    free(h);
    return 0;
}
