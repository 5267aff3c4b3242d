// lab: cost of the f16 split of 8 samples (round 3, f16 Gaussian route), variants, 1 and 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma_pure(f32x2 a, f32x2 b, f32x2 c) { f32x2 o; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c)); return o; }
__device__ __forceinline__ f32x2 pk_sub_pure(f32x2 a, f32x2 c) { f32x2 o; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(a), "v"(c)); return o; }
template <int V>
__device__ __forceinline__ void split8(const float (&x)[8], float q, float mc, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        f32x2 d, hf, r;
        if (V == 0) {
            d = pk_fma_pure(f32x2{x[2 * u], x[2 * u + 1]}, f32x2{q, q}, f32x2{mc, mc});
            hf = __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2, d) & u32x2{0xFFFFE000u, 0xFFFFE000u});
            r = pk_sub_pure(d, hf);
        } else {
            d[0] = __builtin_fmaf(x[2 * u], q, mc);
            d[1] = __builtin_fmaf(x[2 * u + 1], q, mc);
            hf = __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2, d) & u32x2{0xFFFFE000u, 0xFFFFE000u});
            r[0] = d[0] - hf[0];
            r[1] = d[1] - hf[1];
        }
        f16x2 hh, ll;
        if (V == 2) {  // conversions only where needed: h by bit shuffling (v_perm of the two truncated floats is not a conversion: exponent differs) -> keep cvt
            hh = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(hf[0], hf[1]));
            ll = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(r[0], r[1]));
        } else {
            hh = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(hf[0], hf[1]));
            ll = __builtin_convertvector(r, f16x2);
        }
        hi[2 * u] = hh[0]; hi[2 * u + 1] = hh[1]; lo[2 * u] = ll[0]; lo[2 * u + 1] = ll[1];
    }
}
template <int V>
__global__ void k(float* out, int iters, float q, float mc) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.37f + i;
    f16x8 ah, al;
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            split8<V>(x, q, mc, ah, al);
            // feed the result back so that nothing is hoisted or dropped
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] += 1.0f;
            asm volatile("" : : "v"(ah), "v"(al));
        }
    }
    for (int i = 0; i < 8; ++i) s += x[i] + (float)ah[i] + (float)al[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V>
void run(const char* name, int threads, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        k<V><<<256, threads>>>(out, iters, 0.25f, -3.5f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-34s %d waves/SIMD: %.2f ns per split of 8 samples (+8 adds) per wave-slot\n", name, threads / 256, ms * 1e6 / (iters * 8.0));
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    for (int threads : {256, 512, 768}) {
        run<0>("packed f32 (shipped)", threads, out);
        run<1>("scalar fma / sub", threads, out);
        run<2>("packed, both conversions pkrtz", threads, out);
    }
    return 0;
}
