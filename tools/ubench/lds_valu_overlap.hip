// Microbenchmark: do ds_read_b128 traffic and vector-ALU work of the waves of one CU overlap, or add up?
// The 67-px disc kernels issue, per output row of a wave, 42 ds_read_b128 (1 KiB each) and ~350 VALU
// instructions; the counters of tpi_march_kernel say VALU busy + LDS busy ~= kernel time, as if nothing
// overlapped.  This measures, for 1 / 2 / 3 waves per SIMD:
//   valu   : NV v_add_u32 per iteration, nothing else
//   lds    : NL ds_read_b128 per iteration, nothing else (one wait per iteration)
//   both   : the reads issued first, then the adds (independent of the loaded data), then the wait
//   dep    : the adds consume the data loaded in the PREVIOUS iteration (software pipeline, one wait)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_valu_overlap.hip -o /tmp/lds_valu && /tmp/lds_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NL, int NV>
__global__ __launch_bounds__(768) void k(unsigned* out, int iters, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    for (int i = threadIdx.x; i < 32 * 1024; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every wave reads rows of 1 KiB (64 lanes x 16 B): conflict-free, like the prefix rows of the disc kernels
    const unsigned* base = lds + lane * 4 + (wave & 3) * 256;
    unsigned a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    u32x4 cur[NL], nxt[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) cur[i] = *reinterpret_cast<const u32x4*>(base + (i % 24) * 1024);
    const unsigned b = out[threadIdx.x];
    long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned addr = (unsigned)(size_t)base;
    // two half-iterations per trip with the buffers swapped by name, so that no register is copied
#define RD(buf, i) if (i < NL) asm volatile("ds_read_b128 %0, %1 offset:" #i "*4096" : "=v"(buf[i < NL ? i : 0]) : "v"(addr) : "memory");
#define READS(buf) RD(buf, 0) RD(buf, 1) RD(buf, 2) RD(buf, 3) RD(buf, 4) RD(buf, 5) RD(buf, 6) RD(buf, 7) RD(buf, 8) RD(buf, 9) RD(buf, 10) RD(buf, 11) RD(buf, 12) RD(buf, 13) RD(buf, 14) RD(buf, 15)
#define HALF(ld, use)                                                                                                      \
    {                                                                                                                      \
        if (MODE != 0) { READS(ld) }                                                                                       \
        if (MODE == 0 || MODE == 2) {                                                                                      \
            _Pragma("unroll") for (int i = 0; i < NV; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i % 8]) : "v"(b)); \
        }                                                                                                                  \
        if (MODE == 3) {                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < NV; ++i)                                                                 \
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i % 8]) : "v"(use[(i / 4) % NL][i % 4]));                      \
        }                                                                                                                  \
        if (MODE != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
    }
    for (int it = 0; it < iters; it += 2) {
        HALF(nxt, cur)
        HALF(cur, nxt)
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < NL; ++i) s += cur[i][0] + cur[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NL, int NV>
double run(unsigned* d, long long* dc, int waves) {
    const int blocks = 256, iters = 2000;
    hipFuncSetAttribute((const void*)k<MODE, NL, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NL, NV>), dim3(blocks), dim3(waves * 64), 128 * 1024, 0, d, iters, dc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NL, NV>), dim3(blocks), dim3(waves * 64), 128 * 1024, 0, d, iters, dc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // nanoseconds per iteration of one wave-slot round on a CU (all waves of the block do one iteration each)
    return ms * 1e6 / iters;
}

template <int NL, int NV>
void table(unsigned* d, long long* dc) {
    printf("per iteration and wave: %d ds_read_b128 + %d v_add_u32   (ns per iteration of the whole block, 256 blocks on 256 CUs)\n", NL, NV);
    printf("  waves/CU   valu only   lds only    both (indep)   both (dependent, pipelined)   sum    max\n");
    for (int waves : {4, 8, 12}) {
        const double v = run<0, NL, NV>(d, dc, waves), l = run<1, NL, NV>(d, dc, waves), b = run<2, NL, NV>(d, dc, waves),
                     q = run<3, NL, NV>(d, dc, waves);
        printf("  %5d     %8.1f    %8.1f    %8.1f       %8.1f                 %8.1f %8.1f\n", waves, v, l, b, q, v + l, v > l ? v : l);
    }
}

int main() {
    unsigned* d;
    long long* dc;
    hipMalloc(&d, 256 * 1024 * sizeof(unsigned));
    hipMemset(d, 0, 256 * 1024 * sizeof(unsigned));
    hipMalloc(&dc, 256 * sizeof(long long));
    table<8, 64>(d, dc);
    table<8, 32>(d, dc);
    table<8, 128>(d, dc);
    table<4, 64>(d, dc);
    return 0;
}
