// Microbenchmark: issue rate on gfx950 of the VALU instructions the TPI finalisation is made of -
// the float64 ones it uses today and the int32 / float32 ones an integer finalisation would use.
// 8 independent dependency chains per lane, 4 waves per SIMD, every instruction through asm volatile.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/f64_rate.hip -o tools/ubench/f64_rate.out
#include <hip/hip_runtime.h>
#include <cstdio>

enum Op { ADD_F32, FMA_F32, ADD_F64, MUL_F64, FMA_F64, CVT_F64_I32, CVT_F64_F32, CVT_F32_F64, CVT_F32_I32, MAD_I24, NOPS };
static const char* kNames[NOPS] = {"v_add_f32", "v_fma_f32", "v_add_f64", "v_mul_f64", "v_fma_f64", "v_cvt_f64_i32",
                                   "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cvt_f32_i32", "v_mad_i32_i24"};

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    float f[8];
    double d[8];
    int i[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        f[k] = (float)(threadIdx.x + k);
        d[k] = (double)(threadIdx.x + 2 * k);
        i[k] = (int)threadIdx.x + 3 * k;
    }
    const float xf = out[threadIdx.x] + 1.0f;
    const double xd = (double)xf;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (OP == ADD_F32) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f[k]) : "v"(f[k]), "v"(xf));
            if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(f[k]) : "v"(f[k]), "v"(xf));
            if (OP == ADD_F64) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d[k]) : "v"(d[k]), "v"(xd));
            if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[k]) : "v"(d[k]), "v"(xd));
            if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(d[k]) : "v"(d[k]), "v"(xd));
            if (OP == CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[k]) : "v"(i[k]));
            if (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(f[k]));
            if (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[k]) : "v"(d[k]));
            if (OP == CVT_F32_I32) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[k]) : "v"(i[k]));
            if (OP == MAD_I24) asm volatile("v_mad_i32_i24 %0, %1, %2, %1" : "=v"(i[k]) : "v"(i[k]), "v"(i[(k + 1) & 7]));
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += f[k] + (float)d[k] + (float)i[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
float run(float* d, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int blocks = 256 * 4, iters = 10000;  // 4 blocks of 4 waves per CU: 4 waves per SIMD
    float* d;
    hipMalloc(&d, blocks * 256 * sizeof(float));
    hipMemset(d, 0, blocks * 256 * sizeof(float));
    const double ops = (double)blocks * 256 * iters * 8;
    float t[NOPS] = {run<ADD_F32>(d, blocks, iters), run<FMA_F32>(d, blocks, iters), run<ADD_F64>(d, blocks, iters),
                     run<MUL_F64>(d, blocks, iters), run<FMA_F64>(d, blocks, iters), run<CVT_F64_I32>(d, blocks, iters),
                     run<CVT_F64_F32>(d, blocks, iters), run<CVT_F32_F64>(d, blocks, iters),
                     run<CVT_F32_I32>(d, blocks, iters), run<MAD_I24>(d, blocks, iters)};
    for (int k = 0; k < NOPS; ++k)
        printf("%-14s %8.3f ms  %8.1f Glane-op/s  %5.2fx v_add_f32\n", kNames[k], t[k], ops / t[k] / 1e6, t[k] / t[0]);
    return 0;
}
