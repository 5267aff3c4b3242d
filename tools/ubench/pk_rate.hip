// Microbenchmark: does v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 double the f32 rate per
// instruction on gfx950, or does it issue at half rate?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f2 a[8];
    float s[8];
    for (int i = 0; i < 8; ++i) { a[i] = f2{(float)threadIdx.x + i, 1.0f}; s[i] = (float)threadIdx.x + i; }
    f2 x = f2{out[threadIdx.x], out[threadIdx.x + 1]};
    f2 y = f2{1.0001f, 0.9999f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
            if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(x.x), "v"(y.x));
            if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
            if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
            if (MODE == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(x.x), "v"(y.x));
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y + s[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
float run(float* d, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    const int blocks = 256 * 8, iters = 20000;
    float* d; hipMalloc(&d, (blocks * 256 + 8) * sizeof(float)); hipMemset(d, 0, (blocks * 256 + 8) * sizeof(float));
    const double instr = (double)blocks * 256 * iters * 8;
    const char* names[] = {"v_pk_fma_f32", "v_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_max3_f32"};
    float t[5] = {run<0>(d, blocks, iters), run<1>(d, blocks, iters), run<2>(d, blocks, iters), run<3>(d, blocks, iters), run<4>(d, blocks, iters)};
    for (int i = 0; i < 5; ++i) printf("%-14s %8.3f ms  %7.1f G lane-instr/s\n", names[i], t[i], instr / t[i] / 1e6);
    return 0;
}
