// Microbenchmark: issue cost (cycles per wave-instruction per SIMD) of the integer / DPP instructions
// the disc chains are made of, at the occupancy the kernels run at (1 block of 12 waves per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_mix.hip -o /tmp/valu_mix && /tmp/valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters, long long* cyc) {
    extern __shared__ unsigned lds[];
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned b = out[threadIdx.x], c = b + 7;
    lds[threadIdx.x] = b;
    __syncthreads();
    unsigned la = (threadIdx.x * 16) & 0xffff;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define OPS(INS)                                                                                                  \
    asm volatile(REP8(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7))                                      \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                     \
                 : "v"(b), "v"(c), "v"(la)                                                                        \
                 : "vcc", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "s4");
#define I_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_SUB(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_MOVSHL(n) "v_mov_b32_dpp %" #n ", %" #n " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_MOVSHR(n) "v_mov_b32_dpp %" #n ", %" #n " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_ADDSHL(n) "v_add_u32_dpp %" #n ", %" #n ", %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_MOVROW(n) "v_mov_b32_dpp %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_ADDROW(n) "v_add_u32_dpp %" #n ", %" #n ", %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_MOVQUAD(n) "v_mov_b32_dpp %" #n ", %" #n " quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n"
#define I_MOVROL(n) "v_mov_b32_dpp %" #n ", %" #n " wave_rol:1 row_mask:0xf bank_mask:0xf\n"
#define I_MOVBC(n) "v_mov_b32_dpp %" #n ", %" #n " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_F32(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_CVT(n) "v_cvt_f64_i32 v[200:201], %" #n "\n"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 3, %8\n"
#define I_MAXU(n) "v_max_u32 %" #n ", %" #n ", %8\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CMP(n) "v_cmp_neq_f32 vcc, %" #n ", %8\n"
#define I_CVTI(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define I_CVTF(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define I_CVTD(n) "v_cvt_f64_i32 v[200:201], %" #n "\n"
#define I_CVTDF(n) "v_cvt_f64_f32 v[200:201], %" #n "\n"
#define I_CVTFD(n) "v_cvt_f32_f64 %" #n ", v[202:203]\n"
#define I_ADDD(n) "v_add_f64 v[200:201], v[202:203], v[204:205]\n"
#define I_MULD(n) "v_mul_f64 v[200:201], v[202:203], v[204:205]\n"
#define I_FMAD(n) "v_fma_f64 v[200:201], v[202:203], v[204:205], v[206:207]\n"
#define I_MAD64(n) "v_mad_u64_u32 v[200:201], vcc, %" #n ", %8, v[202:203]\n"
#define I_LSHLADD64(n) "v_lshl_add_u64 v[200:201], v[202:203], 2, v[204:205]\n"
#define I_SUBS(n) "v_sub_u32 %" #n ", s4, %" #n "\n"
#define I_MAX3(n) "v_max3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_MAXF(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define I_MINF(n) "v_min_f32 %" #n ", %" #n ", %8\n"
#define I_MAX3F(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MULF(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_SUBF(n) "v_sub_f32 %" #n ", %" #n ", %8\n"
#define I_PKFMA(n) "v_pk_fma_f32 v[200:201], v[202:203], v[204:205], v[206:207]\n"
#define I_MIX(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\nv_sub_u32 %" #n ", %" #n ", %8\n"
        if (MODE == 0) { OPS(I_ADD) }
        if (MODE == 1) { OPS(I_SUB) }
        if (MODE == 2) { OPS(I_ADD3) }
        if (MODE == 3) { OPS(I_MOVSHL) }
        if (MODE == 4) { OPS(I_MOVSHR) }
        if (MODE == 5) { OPS(I_ADDSHL) }
        if (MODE == 6) { OPS(I_MOVROW) }
        if (MODE == 7) { OPS(I_ADDROW) }
        if (MODE == 8) { OPS(I_MOVQUAD) }
        if (MODE == 9) { OPS(I_MOVROL) }
        if (MODE == 10) { OPS(I_MOVBC) }
        if (MODE == 11) { OPS(I_MOV) }
        if (MODE == 12) { OPS(I_F32) }
        if (MODE == 13) { OPS(I_FMA) }
        if (MODE == 14) { OPS(I_LSHLADD) }
        if (MODE == 15) { OPS(I_MAXU) }
        if (MODE == 16) { OPS(I_AND) }
        if (MODE == 17) { OPS(I_CNDMASK) }
        if (MODE == 18) { OPS(I_CMP) }
        if (MODE == 19) { OPS(I_CVTI) }
        if (MODE == 20) { OPS(I_CVTF) }
        if (MODE == 21) { OPS(I_CVTD) }
        if (MODE == 22) { OPS(I_CVTDF) }
        if (MODE == 23) { OPS(I_CVTFD) }
        if (MODE == 24) { OPS(I_ADDD) }
        if (MODE == 25) { OPS(I_MULD) }
        if (MODE == 26) { OPS(I_FMAD) }
        if (MODE == 27) { OPS(I_MAD64) }
        if (MODE == 28) { OPS(I_LSHLADD64) }
        if (MODE == 29) { OPS(I_SUBS) }
        if (MODE == 30) { OPS(I_MAX3) }
        if (MODE == 31) { OPS(I_MAXF) }
        if (MODE == 32) { OPS(I_MAX3F) }
        if (MODE == 33) { OPS(I_MULF) }
        if (MODE == 34) { OPS(I_SUBF) }
        if (MODE == 35) { OPS(I_PKFMA) }
        if (MODE == 36) { OPS(I_MINF) }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, unsigned* d, long long* dc, int waves) {
    const int blocks = 256, iters = 2000;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(waves * 64), 100 * 1024, 0, d, iters, dc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(waves * 64), 100 * 1024, 0, d, iters, dc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), dc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= blocks;
    // per SIMD: waves / 4 waves, each iters * 64 instructions
    const double instr_per_simd = (double)waves / 4.0 * iters * 64.0;
    printf("%-34s %2d waves/CU: %8.3f ms  %6.2f cyc/instr/SIMD (s_memtime)  %6.2f ns/instr/SIMD (wall)\n", name, waves, ms,
           avg / instr_per_simd, ms * 1e6 / instr_per_simd);
}

int main() {
    unsigned* d;
    long long* dc;
    hipMalloc(&d, 256 * 1024 * sizeof(unsigned));
    hipMemset(d, 0, 256 * 1024 * sizeof(unsigned));
    hipMalloc(&dc, 256 * sizeof(long long));
    for (int waves : {8, 12, 16}) {
        run<0>("v_add_u32", d, dc, waves);
        run<1>("v_sub_u32", d, dc, waves);
        run<2>("v_add3_u32", d, dc, waves);
        run<3>("v_mov_b32_dpp wave_shl:1", d, dc, waves);
        run<4>("v_mov_b32_dpp wave_shr:1", d, dc, waves);
        run<5>("v_add_u32_dpp wave_shl:1", d, dc, waves);
        run<6>("v_mov_b32_dpp row_shr:1", d, dc, waves);
        run<7>("v_add_u32_dpp row_shr:1", d, dc, waves);
        run<8>("v_mov_b32_dpp quad_perm", d, dc, waves);
        run<9>("v_mov_b32_dpp wave_rol:1", d, dc, waves);
        run<10>("v_mov_b32_dpp row_bcast:15", d, dc, waves);
        run<11>("v_mov_b32", d, dc, waves);
        run<12>("v_add_f32", d, dc, waves);
        run<13>("v_fma_f32", d, dc, waves);
        if (waves == 8 || waves == 16) {
            run<14>("v_lshl_add_u32", d, dc, waves);
            run<15>("v_max_u32", d, dc, waves);
            run<16>("v_and_b32", d, dc, waves);
            run<17>("v_cndmask_b32", d, dc, waves);
            run<18>("v_cmp_neq_f32", d, dc, waves);
            run<19>("v_cvt_i32_f32", d, dc, waves);
            run<20>("v_cvt_f32_i32", d, dc, waves);
            run<21>("v_cvt_f64_i32", d, dc, waves);
            run<22>("v_cvt_f64_f32", d, dc, waves);
            run<23>("v_cvt_f32_f64", d, dc, waves);
            run<24>("v_add_f64", d, dc, waves);
            run<25>("v_mul_f64", d, dc, waves);
            run<26>("v_fma_f64", d, dc, waves);
            run<27>("v_mad_u64_u32", d, dc, waves);
            run<28>("v_lshl_add_u64", d, dc, waves);
            run<29>("v_sub_u32 sgpr", d, dc, waves);
            run<30>("v_max3_u32", d, dc, waves);
            run<31>("v_max_f32", d, dc, waves);
            run<32>("v_max3_f32", d, dc, waves);
            run<33>("v_mul_f32", d, dc, waves);
            run<34>("v_sub_f32", d, dc, waves);
            run<35>("v_pk_fma_f32", d, dc, waves);
            run<36>("v_min_f32", d, dc, waves);
        }
    }
    return 0;
}
