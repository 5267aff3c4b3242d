// Lab bench for the 67-px TPI / STD kernels: builds variants of the disc kernels from the product's headers in
// one translation unit (30-60 s instead of the whole library), times them on the bench DEM and compares the
// output plane with the product library's, bit for bit.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DMARCH_STAMPS] tools/ubench/tpi_lab.hip \
//         -Ltopo_descriptors_amd -ltopo_amd -Wl,-rpath,'$ORIGIN/../../topo_descriptors_amd' -o tools/ubench/tpi_lab.bin
//   tools/ubench/tpi_lab.bin <variant> [n=32768] [reps=5] [integer=1]
//     variant march : tpi_march_kernel<67, 60, 12> as the product launches it (+ the two empty follow-up launches)
//     variant ring  : tpi_ring_kernel<67, 8> (round 2, 8 columns per lane)
//     variants any_tpi / any_std / any_tpi_std : the product's dispatch for 67 px, compiled with this build's flags
//   (-DMARCH_DYN_ROWS=0/1, -DCHAIN_PRIO=0/1, -DMARCH_STAMPS)
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/topo_amd.h"
#include "../../topo_descriptors_amd/csrc/disc_wave_impl.hpp"
#include "../../topo_descriptors_amd/csrc/disc_ring_impl.hpp"
#ifdef LAB_EXTRA
#include LAB_EXTRA
#endif

#ifndef LAB_TH
#define LAB_TH 60  // rows a marching tile adds (12 waves: a multiple of 12)
#endif

#define CK(x)                                                                         \
    do {                                                                              \
        int rc_ = (x);                                                                \
        if (rc_ != TOPO_AMD_OK) {                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #x, rc_, topo_amd_last_error());        \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

__global__ void diff_kernel(const uint32_t* a, const uint32_t* b, size_t n, unsigned long long* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) bad += a[i] != b[i];
    if (bad) atomicAdd(out, bad);
}

int main(int argc, char** argv) {
    const char* variant = argc > 1 ? argv[1] : "march";
    const int n = argc > 2 ? atoi(argv[2]) : 32768;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    const int integer = argc > 4 ? atoi(argv[4]) : 1;
#ifndef LAB_SIZE
#define LAB_SIZE 67
#endif
    constexpr int SIZE = LAB_SIZE;
    CK(topo_amd_init(0));
    void *dem = nullptr, *out = nullptr, *ref = nullptr, *out2 = nullptr, *ref2 = nullptr;
    const size_t bytes = (size_t)n * n * 4;
    CK(topo_amd_malloc(&dem, bytes));
    CK(topo_amd_malloc(&out, bytes));
    CK(topo_amd_malloc(&ref, bytes));
    CK(topo_amd_synth_dem_dev((float*)dem, n, 0, n, 0, integer));
    topo::Block b{(const float*)dem, n, 0, n, n, 0, n};
    const bool want_std = strstr(variant, "std") != nullptr;
    if (want_std) {
        CK(topo_amd_malloc(&out2, bytes));
        CK(topo_amd_malloc(&ref2, bytes));
    }
    // the product's answer
    CK(topo_amd_tpi_std_dev((const float*)dem, n, 0, n, n, SIZE, 0, n, (float*)ref, (float*)ref2));
    if (!strcmp(variant, "any_std")) CK(topo_amd_memcpy_d2d(out, ref, bytes));  // no TPI plane in this variant
    CK(topo_amd_memset(out, 0xFF, bytes));
    if (out2) CK(topo_amd_memset(out2, 0xFF, bytes));
    auto run = [&]() -> int {
        if (!strcmp(variant, "march")) {
            int rc = topo::launch_march<SIZE, LAB_TH, 12, true, true, true>(b, (float*)out);
            if (rc != TOPO_AMD_OK) return rc;
            rc = topo::launch_fraction_march<SIZE, LAB_TH, 12>(b, (float*)out);
            if (rc != TOPO_AMD_OK) return rc;
            // (the general kernel's tile of LAB_TH > 60 rows does not fit LDS; on the bench DEM nothing is deferred to it)
            if constexpr (LAB_TH <= 60) return topo::launch_wave<SIZE, LAB_TH, 12, true, false>(b, (float*)out, nullptr, true);
            return TOPO_AMD_OK;
        }
        if (!strcmp(variant, "ring")) return topo::launch_ring<SIZE, 8, topo::kRingMain>(b, (float*)out);
        // the product's dispatch (launch_wave_any) with this build's -D flags
        if (!strcmp(variant, "any_tpi")) return topo::launch_wave_any<SIZE>(b, (float*)out, nullptr);
        if (!strcmp(variant, "any_std")) return topo::launch_wave_any<SIZE>(b, nullptr, (float*)out2);
        if (!strcmp(variant, "any_tpi_std")) return topo::launch_wave_any<SIZE>(b, (float*)out, (float*)out2);
#ifdef LAB_EXTRA
        return lab_run(variant, b, (float*)out, (float*)out2);
#else
        topo::set_error("unknown variant %s", variant);
        return TOPO_AMD_EINVAL;
#endif
    };
    CK(run());
    CK(topo_amd_sync());
    float best = 1e30f, sum = 0.f;
    for (int r = 0; r < reps; ++r) {
        CK(topo_amd_mark(2 * r));
        CK(run());
        CK(topo_amd_mark(2 * r + 1));
    }
    for (int r = 0; r < reps; ++r) {
        float ms = 0.f;
        CK(topo_amd_mark_elapsed(2 * r, 2 * r + 1, &ms));
        best = ms < best ? ms : best;
        sum += ms;
    }
    unsigned long long* d_bad = nullptr;
    unsigned long long bad[2] = {0, 0};
    CK(topo_amd_malloc((void**)&d_bad, 16));
    CK(topo_amd_memset(d_bad, 0, 16));
    CK(topo_amd_sync());
    hipLaunchKernelGGL(diff_kernel, dim3(4096), dim3(256), 0, 0, (const uint32_t*)out, (const uint32_t*)ref, (size_t)n * n, d_bad);
    if (out2) hipLaunchKernelGGL(diff_kernel, dim3(4096), dim3(256), 0, 0, (const uint32_t*)out2, (const uint32_t*)ref2, (size_t)n * n, d_bad + 1);
    (void)hipDeviceSynchronize();
    CK(topo_amd_memcpy_d2h(bad, d_bad, 16));
    printf("{\"variant\": \"%s\", \"integer_dem\": %d, \"n\": %d, \"ms_mean\": %.4f, \"ms_min\": %.4f, \"reps\": %d, "
           "\"pixels_differing_from_product\": %llu, \"std_pixels_differing\": %llu}\n",
           variant, integer, n, sum / reps, best, reps, bad[0], bad[1]);
    return 0;
}
