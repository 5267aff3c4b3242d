// Microbenchmark: cycles per v_mfma_f32_32x32x2_f32 for one wave per SIMD (256-thread blocks, one per CU),
// the shape the matrix-core Gaussian kernels run in.
//   mode 0: one dependent chain (same accumulator), nothing else
//   mode 1: two independent chains, alternating
//   mode 2: one chain, one v_sub_f32 before each MFMA (the accumulation offset)
//   mode 3: one chain, v_sub + a ds_read2_b32 pair per MFMA (operands one phase ahead)
//   mode 4: one chain, 4 waves per SIMD (1024-thread blocks)
//   mode 5: one chain + a v_sub per MFMA whose result no MFMA reads
//   mode 6: one chain + a v_sub per MFMA computed two MFMAs ahead of its use
//   mode 7: one chain + one v_pk_add_f32 per two MFMAs
//   mode 8: 8 v_sub in a burst, then 8 MFMAs
//   mode 9: one chain + an s_add per MFMA (scalar ALU)
//   mode 10: one chain, operands rotating through 16 registers of random data (power: the constant operands of
//            the other modes toggle nothing)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_chain.hip -o tools/ubench/mfma_chain.bin && tools/ubench/mfma_chain.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void k(float* out, int iters, long long* cyc) {
    __shared__ float lds[4096];
    for (int n = threadIdx.x; n < 4096; n += blockDim.x) lds[n] = 1.0f;
    __syncthreads();
    f32x16 acc0, acc1;
    for (int v = 0; v < 16; ++v) acc0[v] = acc1[v] = 0.0f;
    float a = out[threadIdx.x], b = a + 1.0f, c = 0.5f;
    const float* p = lds + (threadIdx.x & 63);
    float r0 = a, r1 = b, r2 = a, r3 = b;
    float rnd[16];
    {
        unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int n = 0; n < 16; ++n) {
            x = x * 1664525u + 1013904223u;
            rnd[n] = (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f + (MODE == 10 ? 0.0f : a);
        }
    }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0 || MODE == 4) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            } else if (MODE == 1) {
                if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            } else if (MODE == 2) {
                float s;
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s) : "v"(a), "v"(c));
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, b, acc0, 0, 0, 0);
            } else if (MODE == 3) {
                float s;
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s) : "v"(r0), "v"(c));
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, r1, acc0, 0, 0, 0);
                r0 = r2;
                r1 = r3;
                r2 = p[64 * u];
                r3 = p[64 * u + 2048];
            } else if (MODE == 5) {
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r2) : "v"(r2), "v"(c));
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            } else if (MODE == 6) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(r0, b, acc0, 0, 0, 0);
                r0 = r1;
                r1 = r2;
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r2) : "v"(a), "v"(c));
            } else if (MODE == 7) {
                if ((u & 1) == 0) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 in = {a, b}, cc = {c, c}, o;
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(o) : "v"(in), "v"(cc));
                    r0 = o[0];
                    r1 = o[1];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(r0, b, acc0, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(r1, b, acc0, 0, 0, 0);
                }
            } else if (MODE == 10) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(rnd[2 * u], rnd[2 * u + 1], acc0, 0, 0, 0);
            } else if (MODE == 9) {
                asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            }
        }
        if (MODE == 8) {
            float s[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s[u]) : "v"(a), "v"(c));
#pragma unroll
            for (int u = 0; u < 8; ++u) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(s[u], b, acc0, 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int v = 0; v < 16; ++v) s += acc0[v] + acc1[v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + r0 + r1 + r2 + rnd[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads) {
    const int blocks = 256, iters = 20000;
    float* out;
    long long* cyc;
    hipMalloc(&out, blocks * 1024 * sizeof(float));
    hipMemset(out, 0, blocks * 1024 * sizeof(float));
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 100, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)iters * 8 * (threads / 256);
    // s_memtime ticks at 100 MHz on this part: report the wall-clock figure (ns per MFMA per SIMD)
    printf("%-44s %7.3f ms  %6.2f ns per MFMA per SIMD  (%5.1f cycles at 2.4 GHz; memtime %lld)\n", name, ms,
           ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, c);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("one chain", 256);
    run<1>("two chains alternating", 256);
    run<2>("one chain + v_sub", 256);
    run<3>("one chain + v_sub + 2 ds_read", 256);
    run<4>("one chain, 4 waves per SIMD", 1024);
    run<5>("one chain + independent v_sub", 256);
    run<6>("one chain + v_sub two MFMAs ahead", 256);
    run<7>("one chain + v_pk_add per two MFMAs", 256);
    run<8>("8 v_sub burst, then 8 MFMAs", 256);
    run<9>("one chain + s_add", 256);
    run<10>("one chain, random operands", 256);
    run<0>("one chain (again)", 256);
    return 0;
}
