// lab: what build_tap_blocks-style code and the f16 MFMA really compute (round 3, f16 Gaussian route)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* taps, int n, float scale, float* hi, float* lo, float* prod) {
    const int lane = threadIdx.x;
    f16x8 h, l;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int i = 8 * (lane >> 5) + q;
        const float t = i < n ? taps[i] * scale : 0.0f;
        const _Float16 th = (_Float16)t;
        h[q] = th;
        l[q] = (_Float16)(t - (float)th);
    }
    for (int q = 0; q < 8; ++q) { hi[lane * 8 + q] = (float)h[q]; lo[lane * 8 + q] = (float)l[q]; }
    // product test: A = taps (every row the same 16 taps), B = identity-like: column n has 1 at k = n & 15
    f16x8 b;
    for (int q = 0; q < 8; ++q) b[q] = (8 * (lane >> 5) + q == (lane & 15)) ? (_Float16)1.0f : (_Float16)0.0f;
    f32x16 acc;
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(l, b, acc, 0, 0, 0);
    for (int v = 0; v < 16; ++v) prod[lane * 16 + v] = acc[v];
}
int main() {
    std::vector<float> taps(16);
    for (int i = 0; i < 16; ++i) taps[i] = std::exp(-0.5f * (i - 8) * (i - 8) / 5.0625f) * 0.1773f;
    float *d, *hi, *lo, *pr;
    (void)hipMalloc(&d, 64); (void)hipMalloc(&hi, 64 * 8 * 4); (void)hipMalloc(&lo, 64 * 8 * 4); (void)hipMalloc(&pr, 64 * 16 * 4);
    (void)hipMemcpy(d, taps.data(), 64, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, 16, 4096.f, hi, lo, pr);
    std::vector<float> H(512), Lo(512), P(1024);
    (void)hipMemcpy(H.data(), hi, 2048, hipMemcpyDeviceToHost); (void)hipMemcpy(Lo.data(), lo, 2048, hipMemcpyDeviceToHost); (void)hipMemcpy(P.data(), pr, 4096, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) {
        const float t = taps[i] * 4096.f;
        const int lane = (i >> 3) * 32, q = i & 7;
        printf("tap %2d t %.6f hi %.6f lo %.9f  t-hi-lo %.3e   mfma(lo x e_k) column %d row 0: %.9f\n", i, t, H[lane * 8 + q], Lo[lane * 8 + q], t - H[lane * 8 + q] - Lo[lane * 8 + q], i, P[i * 16 + 0]);
    }
    return 0;
}
