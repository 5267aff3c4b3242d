// lab: cycles per v_mfma_f32_32x32x16_f16 with 1, 2 and 4 independent accumulators, one wave per SIMD (round 3)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* clk) {
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(threadIdx.x * 0.001f + q); b[q] = (_Float16)(q * 0.5f); }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int v = 0; v < 16; ++v) acc[n][v] = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24 / NACC; ++u)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int v = 0; v < 16; ++v) s += acc[n][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}
int main() {
    float* out; long long* clk;
    (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&clk, 8);
    const int iters = 2000;
    for (int nacc : {1, 2, 3, 4}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (nacc == 1) k<1><<<256, 256>>>(out, iters, clk);
            if (nacc == 2) k<2><<<256, 256>>>(out, iters, clk);
            if (nacc == 3) k<3><<<256, 256>>>(out, iters, clk);
            if (nacc == 4) k<4><<<256, 256>>>(out, iters, clk);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long c; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("accumulators %d: %.3f ms for %d MFMAs per wave -> %.1f ns per MFMA, s_memtime %.1f ticks per MFMA (100 MHz ticks)\n", nacc, ms, iters * 24,
               ms * 1e6 / (iters * 24.0), (double)c / (iters * 24.0));
    }
    return 0;
}
