// lab: operand and result layout of v_mfma_f32_16x16x32_f16 (round 3, tall tiles of the f16 Gaussian)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* C) {  // A 16 x 32, B 32 x 16 row-major, C 16 x 16
    const int lane = threadIdx.x, m = lane & 15, kg = lane >> 4;
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)A[m * 32 + 8 * kg + q]; b[q] = (_Float16)B[(8 * kg + q) * 16 + m]; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) C[(4 * kg + v) * 16 + m] = acc[v];  // assumed: lane -> column m, rows 4 kg + v
}
int main() {
    std::vector<float> A(512), B(512), C(256), R(256, 0.f);
    for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7) % 13 - 6); B[i] = (float)((i * 5) % 11 - 5); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int kk = 0; kk < 32; ++kk) R[i * 16 + j] += A[i * 32 + kk] * B[kk * 16 + j];
    float *dA, *dB, *dC;
    (void)hipMalloc(&dA, 2048); (void)hipMalloc(&dB, 2048); (void)hipMalloc(&dC, 1024);
    (void)hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC);
    (void)hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += C[i] != R[i];
    printf("16x16x32 f16: A lane (m = l & 15, k = 8 (l >> 4) + q), B lane (n = l & 15, same k), D lane (n = l & 15, rows 4 (l >> 4) + v): %d mismatches of 256\n", bad);
    return 0;
}
