// A/B for the non-output write traffic of the exact one-pass TPI kernel (67 px, integer DEM).
//
//   variant 12: disc_wave_kernel<67, 60, 12, true, false>  - the product's choice; 168-VGPR budget,
//               22 VGPRs spilled to scratch (8 dword stores per lane per tile inside the tile loop)
//   variant  8: disc_wave_kernel<67, 64,  8, true, false>  - 256-VGPR budget
//   variant  0: the product's pair - the marching build (tpi_march_kernel: integer tiles only, carries
//               the window down a column strip), then the general 12-wave build over the deferred tiles
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
        // variant 0: what the product launches - marching build, then the general build over the deferred tiles
        int rc = topo::launch_march<67, 60, 12, true, true, true>(b, (float*)out);
        if (rc != TOPO_AMD_OK) return rc;
        rc = topo::launch_fraction_march<67, 60, 12>(b, (float*)out);
        if (rc != TOPO_AMD_OK) return rc;
        return topo::launch_wave<67, 60, 12, true, false>(b, (float*)out, nullptr, true);
    };
    CK(run());
    CK(topo_amd_sync());
    // back-to-back: one event pair around all launches (what bench.py does)
    float ms = 0.f;
    CK(topo_amd_timer_start());
    for (int r = 0; r < reps; ++r) CK(run());
    CK(topo_amd_timer_stop(&ms));
    // isolated: one event pair and a synchronise per launch
    float iso_min = 1e30f, iso_max = 0.f, iso_sum = 0.f;
    for (int r = 0; r < reps; ++r) {
        float one = 0.f;
        CK(topo_amd_timer_start());
        CK(run());
        CK(topo_amd_timer_stop(&one));
        iso_min = one < iso_min ? one : iso_min;
        iso_max = one > iso_max ? one : iso_max;
        iso_sum += one;
    }
    // back-to-back once more, after the isolated launches
    float ms2 = 0.f;
    CK(topo_amd_timer_start());
    for (int r = 0; r < reps; ++r) CK(run());
    CK(topo_amd_timer_stop(&ms2));
    // checksum so the two variants can be compared for identical output
    const size_t probe = (size_t)n * 1024;
    float* h = (float*)malloc(probe * 4);
    CK(topo_amd_memcpy_d2h(h, (const char*)out + ((size_t)n / 2) * n * 4, probe * 4));
    double s = 0.0;
    for (size_t i = 0; i < probe; ++i) s += (double)h[i] * (double)((i % 97) + 1);
    printf("{\"variant\": %d, \"integer_dem\": %d, \"n\": %d, \"ms\": %.4f, \"ms_isolated_mean\": %.4f, \"ms_isolated_min\": %.4f, \"ms_isolated_max\": %.4f, \"ms_back_to_back_again\": %.4f, \"reps\": %d, \"output_bytes\": %zu, \"checksum\": %.6f}\n", variant, integer, n,
           ms / reps, iso_sum / reps, iso_min, iso_max, ms2 / reps, reps, (size_t)n * n * 4, s);
    free(h);
    return 0;
}
