// Lab (round 6, profiles/r06_gauss_axis0_s1.txt): what does the memory system give a kernel with the access pattern of the Gaussian's
// axis-0 passes - one block per 128-column strip marching down the rows, reading and writing 32 rows x 512 bytes per step - when the
// kernel does NOTHING else?  usage: strip_copy [n=32768] [mode: 0 strips read+write, 1 strips read only, 2 linear read+write, 3 strips, 4 strips per CU 256 cols]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int COLS, bool WRITE>
__global__ __launch_bounds__(256) void strip_kernel(const float* in, float* out, int nx, int ny) {
    const int x0 = blockIdx.x * COLS;
    constexpr int TPR = COLS / 4, RPI = 256 / TPR;  // threads per row, rows per instruction
    const int r = threadIdx.x / TPR, c = (threadIdx.x % TPR) * 4;
    f4 keep = {0, 0, 0, 0};
    for (int y = 0; y < ny; y += 32) {
        f4 v[32 / RPI];
#pragma unroll
        for (int k = 0; k < 32 / RPI; ++k) v[k] = *reinterpret_cast<const f4*>(in + (size_t)(y + r + RPI * k) * nx + x0 + c);
#pragma unroll
        for (int k = 0; k < 32 / RPI; ++k) {
            if (WRITE) *reinterpret_cast<f4*>(out + (size_t)(y + r + RPI * k) * nx + x0 + c) = v[k] + 1.0f;
            else keep += v[k];
        }
    }
    if (!WRITE && keep[0] == 12345.678f) out[0] = keep[1];
}
__global__ __launch_bounds__(256) void linear_kernel(const float* in, float* out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        reinterpret_cast<f4*>(out)[i] = reinterpret_cast<const f4*>(in)[i] + 1.0f;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 32768;
    float *in, *out;
    hipMalloc(&in, (size_t)n * n * 4);
    hipMalloc(&out, (size_t)n * n * 4);
    hipMemset(in, 0, (size_t)n * n * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int mode = 0; mode < 4; ++mode) {
        std::vector<float> ms;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL((strip_kernel<128, true>), dim3(n / 128), dim3(256), 0, 0, in, out, n, n);
            if (mode == 1) hipLaunchKernelGGL((strip_kernel<128, false>), dim3(n / 128), dim3(256), 0, 0, in, out, n, n);
            if (mode == 2) hipLaunchKernelGGL(linear_kernel, dim3(256 * 8), dim3(256), 0, 0, in, out, (size_t)n * n / 4);
            if (mode == 3) hipLaunchKernelGGL((strip_kernel<256, true>), dim3(n / 256), dim3(256), 0, 0, in, out, n, n);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float t;
            hipEventElapsedTime(&t, a, b);
            ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const char* name[] = {"128-column strips, read + write", "128-column strips, read only", "linear grid-stride copy", "256-column strips (128 blocks), read + write"};
        const double bytes = (double)n * n * 4 * (mode == 1 ? 1 : 2);
        printf("%-48s %.3f ms  %.2f TB/s\n", name[mode], ms[2], bytes / ms[2] / 1e9);
    }
    return 0;
}
