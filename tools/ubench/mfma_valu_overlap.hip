// lab: does a wave, alone on its SIMD, run its own vector-ALU instructions under its own MFMAs?  A loop of 8
// v_mfma_f32_32x32x16_f16 (one dependent chain) with K independent v_fma_f32 between consecutive MFMAs, K = 0 ... 12:
// cycles per iteration (s_memtime), 1 or 2 waves per SIMD.  Also the same with ds_read_b128 / global stores in between.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int K, int MODE>
__global__ void k(float* out, unsigned long long* t, int iters) {
    __shared__ float lds[4096];
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(threadIdx.x * 0.01f + q); b[q] = (_Float16)(q * 0.5f); }
    f32x16 acc;
    for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
    float x[12];
    for (int q = 0; q < 12; ++q) x[q] = threadIdx.x + q;
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float ld = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            if (MODE == 0) {
#pragma unroll
                for (int q = 0; q < K; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[q]) : "v"(x[(q + 1) % 12]));
            } else if (MODE == 1) {
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    f4 v;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(((threadIdx.x * 4 + 64 * q + 4 * m) & 4092) * 4)));
                    asm volatile("s_waitcnt lgkmcnt(8)");
                    ld += 0.0f;
                    asm volatile("" ::"v"(v));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = ld;
    for (int v = 0; v < 16; ++v) s += acc[v];
    for (int q = 0; q < 12; ++q) s += x[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t1 - t0;
}
template <int K, int MODE>
void run(int threads, const char* what) {
    float* out; unsigned long long* t;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&t, 8);
    const int iters = 2000;
    k<K, MODE><<<256, threads>>>(out, t, iters);
    k<K, MODE><<<256, threads>>>(out, t, iters);
    unsigned long long h = 0;
    (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%s K=%2d threads=%d: %.1f cycles per MFMA slot\n", what, K, threads, (double)h / iters / 8);
    (void)hipFree(out); (void)hipFree(t);
}
int main() {
    for (int threads : {256, 512}) {
        run<0, 0>(threads, "valu"); run<2, 0>(threads, "valu"); run<4, 0>(threads, "valu"); run<6, 0>(threads, "valu");
        run<8, 0>(threads, "valu"); run<10, 0>(threads, "valu"); run<12, 0>(threads, "valu");
        run<1, 1>(threads, "ds128"); run<2, 1>(threads, "ds128"); run<4, 1>(threads, "ds128");
    }
    return 0;
}
