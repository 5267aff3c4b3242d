// Microbenchmark, second batch: issue cost of the 16-bit packed, dot and SDWA forms a 16-bit disc chain would use
// (column sums of <= 67 samples of |u| <= 489 fit int16: prefix rows of half the LDS bytes, packed subtractions),
// at 8 / 12 / 16 waves per CU like tools/ubench/valu_mix.hip.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_mix2.hip -o /tmp/valu_mix2 && /tmp/valu_mix2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters, long long* cyc) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned b = out[threadIdx.x], c = b + 7;
    const unsigned sel = __builtin_amdgcn_readfirstlane(out[0] | 0x00010001u);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define OPS(INS)                                                                                \
    asm volatile(REP8(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7))                    \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   \
                 : "v"(b), "v"(c), "s"(sel)                                                     \
                 : "vcc");
#define I_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_PKADD16(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n"
#define I_PKSUB16(n) "v_pk_sub_u16 %" #n ", %" #n ", %8\n"
#define I_PKSUBI16(n) "v_pk_sub_i16 %" #n ", %" #n ", %8\n"
#define I_DOT2(n) "v_dot2_i32_i16 %" #n ", %8, %10, %" #n "\n"
#define I_DOT2C(n) "v_dot2c_i32_i16 %" #n ", %8, %9\n"
#define I_DOT4(n) "v_dot4_i32_i8 %" #n ", %8, %10, %" #n "\n"
#define I_SDWA(n) "v_add_u32_sdwa %" #n ", %" #n ", sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define I_SDWA0(n) "v_add_u32_sdwa %" #n ", %" #n ", sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
#define I_BFE(n) "v_bfe_i32 %" #n ", %" #n ", 16, 16\n"
#define I_ASHR(n) "v_ashrrev_i32 %" #n ", 16, %" #n "\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define I_MADI16(n) "v_mad_i32_i16 %" #n ", %8, %9, %" #n "\n"
#define I_ADDSHL(n) "v_add_u32_dpp %" #n ", %" #n ", %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_DOT2SHL(n) "v_dot2c_i32_i16_dpp %" #n ", %8, %9 wave_shl:1 row_mask:0xf bank_mask:0xf\n"
#define I_MADU24(n) "v_mad_u32_u24 %" #n ", %8, %9, %" #n "\n"
#define I_SAD(n) "v_sad_u32 %" #n ", %8, %9, %" #n "\n"
        if (MODE == 0) { OPS(I_ADD) }
        if (MODE == 1) { OPS(I_PKADD16) }
        if (MODE == 2) { OPS(I_PKSUB16) }
        if (MODE == 3) { OPS(I_PKSUBI16) }
        if (MODE == 4) { OPS(I_DOT2) }
        if (MODE == 5) { OPS(I_DOT2C) }
        if (MODE == 6) { OPS(I_DOT4) }
        if (MODE == 7) { OPS(I_SDWA) }
        if (MODE == 8) { OPS(I_SDWA0) }
        if (MODE == 9) { OPS(I_BFE) }
        if (MODE == 10) { OPS(I_ASHR) }
        if (MODE == 11) { OPS(I_PERM) }
        if (MODE == 12) { OPS(I_MADI16) }
        if (MODE == 13) { OPS(I_ADDSHL) }
        if (MODE == 14) { OPS(I_MADU24) }
        if (MODE == 15) { OPS(I_SAD) }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, unsigned* d, long long* dc, int waves) {
    const int blocks = 256, iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(waves * 64), 0, 0, d, iters, dc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(waves * 64), 0, 0, d, iters, dc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)waves / 4.0 * iters * 64.0;
    printf("%-34s %2d waves/CU: %8.3f ms  %6.2f ns/instr/SIMD (wall)  = %5.2f cycles at 2.4 GHz\n", name, waves, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

int main() {
    unsigned* d;
    long long* dc;
    (void)hipMalloc(&d, 256 * 1024 * sizeof(unsigned));
    (void)hipMemset(d, 0, 256 * 1024 * sizeof(unsigned));
    (void)hipMalloc(&dc, 256 * sizeof(long long));
    for (int waves : {8, 12, 16}) {
        run<0>("v_add_u32", d, dc, waves);
        run<1>("v_pk_add_u16", d, dc, waves);
        run<2>("v_pk_sub_u16", d, dc, waves);
        run<3>("v_pk_sub_i16", d, dc, waves);
        run<4>("v_dot2_i32_i16 (sgpr weights)", d, dc, waves);
        run<5>("v_dot2c_i32_i16", d, dc, waves);
        run<6>("v_dot4_i32_i8 (sgpr weights)", d, dc, waves);
        run<7>("v_add_u32_sdwa sext WORD_1", d, dc, waves);
        run<8>("v_add_u32_sdwa sext WORD_0", d, dc, waves);
        run<9>("v_bfe_i32", d, dc, waves);
        run<10>("v_ashrrev_i32", d, dc, waves);
        run<11>("v_perm_b32", d, dc, waves);
        run<12>("v_mad_i32_i16", d, dc, waves);
        run<13>("v_add_u32_dpp wave_shl:1", d, dc, waves);
        run<14>("v_mad_u32_u24", d, dc, waves);
        run<15>("v_sad_u32", d, dc, waves);
    }
    return 0;
}
