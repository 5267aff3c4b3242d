// Microbenchmark: issue rate of v_add_f32 with a DPP wave_shr:1 / wave_shl:1 operand vs a plain
// v_add_f32 on gfx950, and a check of what the shift does at the wave edges.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define DPP_WAVE_SHL1 0x130
#define DPP_WAVE_SHR1 0x138

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (float)(threadIdx.x + k);
    float x = out[threadIdx.x];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) {
                a[k] = a[k] + x;
            } else if (MODE == 1) {
                a[k] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[k]), DPP_WAVE_SHR1, 0xf, 0xf, true)) + x;
            } else if (MODE == 2) {
                a[k] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[k]), DPP_WAVE_SHL1, 0xf, 0xf, true)) + x;
            } else if (MODE == 4) {
                // packed add: two floats per lane per instruction
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 t = {a[k], a[(k + 1) & 7]};
                f2 u = {x, x};
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(u));
                a[k] = t.x;
            } else if (MODE == 5) {
                a[k] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[k]), DPP_WAVE_SHR1, 0xf, 0xf, true));
            } else {
                int v = __float_as_int(a[k]);
                v = __builtin_amdgcn_update_dpp(0, v, DPP_WAVE_SHR1, 0xf, 0xf, true) + __float_as_int(x);
                a[k] = __int_as_float(v);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void edge_kernel(int* out) {
    int v = threadIdx.x + 100;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, DPP_WAVE_SHR1, 0xf, 0xf, false);
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, DPP_WAVE_SHL1, 0xf, 0xf, false);
    out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, DPP_WAVE_SHR1, 0xf, 0xf, true);
}

template <int MODE>
float run(float* d, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int blocks = 256 * 8, iters = 20000;
    float* d;
    hipMalloc(&d, blocks * 256 * sizeof(float));
    hipMemset(d, 0, blocks * 256 * sizeof(float));
    const double ops = (double)blocks * 256 * iters * 8;
    float t0 = run<0>(d, blocks, iters), t1 = run<1>(d, blocks, iters), t2 = run<2>(d, blocks, iters),
          t3 = run<3>(d, blocks, iters);
    printf("plain v_add_f32       : %8.3f ms  %7.1f Glane-op/s\n", t0, ops / t0 / 1e6);
    printf("v_add_f32 wave_shr:1  : %8.3f ms  %7.1f Glane-op/s\n", t1, ops / t1 / 1e6);
    printf("v_add_f32 wave_shl:1  : %8.3f ms  %7.1f Glane-op/s\n", t2, ops / t2 / 1e6);
    printf("v_add_u32 wave_shr:1  : %8.3f ms  %7.1f Glane-op/s\n", t3, ops / t3 / 1e6);
    float t4 = run<4>(d, blocks, iters), t5 = run<5>(d, blocks, iters);
    printf("v_pk_add_f32 (instr)  : %8.3f ms  %7.1f Ginstr-lane/s (x2 flops)\n", t4, ops / t4 / 1e6);
    printf("v_mov_b32 wave_shr:1  : %8.3f ms  %7.1f Glane-op/s\n", t5, ops / t5 / 1e6);
    int* di;
    hipMalloc(&di, 192 * sizeof(int));
    hipLaunchKernelGGL(edge_kernel, dim3(1), dim3(64), 0, 0, di);
    std::vector<int> h(192);
    hipMemcpy(h.data(), di, 192 * sizeof(int), hipMemcpyDeviceToHost);
    printf("wave_shr:1 lanes 0,1,15,16,17,31,32,33,63: %d %d %d %d %d %d %d %d %d\n", h[0], h[1], h[15], h[16], h[17], h[31], h[32], h[33], h[63]);
    printf("wave_shl:1 lanes 0,15,16,31,32,62,63: %d %d %d %d %d %d %d\n", h[64], h[64 + 15], h[64 + 16], h[64 + 31], h[64 + 32], h[64 + 62], h[64 + 63]);
    printf("wave_shr:1 bound_ctrl lane 0: %d\n", h[128]);
    return 0;
}
