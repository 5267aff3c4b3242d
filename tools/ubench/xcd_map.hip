// Where do workgroups land?  (1) XCC id and CU id of every block of a 256-block launch that fills the chip (one block per
// CU by LDS).  (2) A persistent launch A of `na` blocks (140 KB of LDS each, spinning for 400 us) in which the blocks
// listed as absent leave at once, and 30 us later, on another stream, a small launch B of `nb` blocks (37 KB of LDS,
// 256 threads): when does B run, and on which XCCs?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/xcd_map.hip -o /tmp/xcd_map && /tmp/xcd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ inline unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
__device__ inline unsigned hw_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }

__global__ void spin_kernel(unsigned* where, long long* t, int ticks, int absent_from, int absent_mask_xcds) {
    extern __shared__ float lds[];
    const int b = blockIdx.x;
    if (threadIdx.x == 0) { where[2 * b] = xcc_id(); where[2 * b + 1] = hw_id(); t[2 * b] = __builtin_amdgcn_s_memrealtime(); }
    if (b >= absent_from && (b & 7) < absent_mask_xcds) { if (threadIdx.x == 0) t[2 * b + 1] = __builtin_amdgcn_s_memrealtime(); return; }
    lds[threadIdx.x] = 1.0f;
    if (threadIdx.x == 0) {
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
        t[2 * b + 1] = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
}
__global__ void small_kernel(unsigned* where, long long* t) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 2.0f;
    if (threadIdx.x == 0) { where[2 * blockIdx.x] = xcc_id(); where[2 * blockIdx.x + 1] = hw_id(); t[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime(); }
    __syncthreads();
    if (threadIdx.x == 0) t[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}

int main() {
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    unsigned *wa, *wb; long long *ta, *tb;
    CK(hipHostMalloc(&wa, 2 * 512 * 4)); CK(hipHostMalloc(&wb, 2 * 64 * 4));
    CK(hipHostMalloc(&ta, 2 * 512 * 8)); CK(hipHostMalloc(&tb, 2 * 64 * 8));
    const size_t lds_a = 140 * 1024, lds_b = 37888;
    CK(hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
    struct Case { int na, absent_from, absent_xcds, nb; const char* what; };
    const Case cases[] = {{256, 1 << 30, 0, 8, "A 256 blocks, none absent, B 8"},
                          {248, 1 << 30, 0, 8, "A 248 blocks, B 8"},
                          {252, 1 << 30, 0, 4, "A 252 blocks, B 4"},
                          {256, 248, 8, 8, "A 256 blocks, 248..255 absent, B 8"},
                          {256, 248, 4, 4, "A 256 blocks, 248..251 absent, B 4"},
                          {256, 0, 0, 1, "(map only)"}};
    for (const Case& cs : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(spin_kernel, dim3(cs.na), dim3(768), lds_a, sa, wa, ta, 40000, cs.absent_from, cs.absent_xcds);
            // B goes in about 30 us later
            CK(hipStreamQuery(sa) == hipErrorNotReady ? hipSuccess : hipSuccess);
            { const auto t0 = clock(); while ((clock() - t0) * 1e6 / CLOCKS_PER_SEC < 30) {} }
            hipLaunchKernelGGL(small_kernel, dim3(cs.nb), dim3(256), lds_b, sb, wb, tb);
            CK(hipDeviceSynchronize());
        }
        long long a0 = ta[0], a1 = 0;
        for (int b = 0; b < cs.na; ++b) { if (ta[2 * b] < a0) a0 = ta[2 * b]; if (ta[2 * b + 1] > a1) a1 = ta[2 * b + 1]; }
        printf("%s: A runs %.1f us;  B blocks start / end after A's start (us), xcc:", cs.what, (a1 - a0) / 100.0);
        for (int b = 0; b < cs.nb; ++b) printf("  [%d] %.1f-%.1f x%u", b, (tb[2 * b] - a0) / 100.0, (tb[2 * b + 1] - a0) / 100.0, wb[2 * b]);
        printf("\n");
    }
    printf("block -> xcc of the 256-block launch: ");
    for (int b = 0; b < 32; ++b) printf("%u", wa[2 * b]);
    printf(" ... ");
    for (int b = 240; b < 256; ++b) printf("%u", wa[2 * b]);
    printf("\n");
    int per_xcc[16] = {0};
    for (int b = 0; b < 256; ++b) per_xcc[wa[2 * b] & 15]++;
    printf("blocks per xcc:");
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
    return 0;
}
