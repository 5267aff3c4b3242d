// Which way does a device-to-host copy go, and how fast?  The pipelined host-buffer path (csrc/capi.hip, run_pipelined) saw
// 64 MB D2H chunks on a copy-only stream run at 29 GB/s where one 1 GiB copy on the compute stream runs at 54 GB/s
// (rocprofv3 --memory-copy-trace).  This bench times the candidates:
//   build: hipcc --offload-arch=gfx950 -O2 -o d2h_paths.bin d2h_paths.hip        run: ./d2h_paths.bin
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                          \
    do {                                                               \
        hipError_t e = (x);                                            \
        if (e != hipSuccess) {                                         \
            std::printf("%s: %s\n", #x, hipGetErrorString(e));         \
            return 1;                                                  \
        }                                                              \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void tiny(int* p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy16(u4* __restrict__ dst, const u4* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u4 v = __builtin_nontemporal_load(src + i);
        __builtin_nontemporal_store(v, dst + i);
    }
}

int main() {
    const size_t bytes = (size_t)1 << 30, chunk = (size_t)64 << 20;
    void *d0 = nullptr, *d1 = nullptr;
    int* flag = nullptr;
    CK(hipMalloc(&d0, bytes));
    CK(hipMalloc(&d1, bytes));
    CK(hipMalloc((void**)&flag, 64));
    CK(hipMemset(d0, 1, bytes));
    CK(hipMemset(d1, 2, bytes));
    void *pin0 = nullptr, *pin1 = nullptr;
    CK(hipHostMalloc(&pin0, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&pin1, bytes, hipHostMallocDefault));
    std::memset(pin0, 3, bytes);
    std::memset(pin1, 4, bytes);
    char* page = (char*)std::malloc(bytes);
    std::memset(page, 5, bytes);
    hipStream_t sa, sb, sc;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    auto rate = [&](double t) { return bytes / t / 1e9; };
    for (int rep = 0; rep < 2; ++rep) {
        double t = now();
        CK(hipMemcpyAsync(pin0, d0, bytes, hipMemcpyDeviceToHost, sa));
        CK(hipStreamSynchronize(sa));
        std::printf("D2H pinned, 1 GiB, idle copy-only stream:            %6.1f GB/s\n", rate(now() - t));
        t = now();
        for (size_t o = 0; o < bytes; o += chunk) CK(hipMemcpyAsync((char*)pin0 + o, (char*)d0 + o, chunk, hipMemcpyDeviceToHost, sa));
        CK(hipStreamSynchronize(sa));
        std::printf("D2H pinned, 16 x 64 MB, copy-only stream:            %6.1f GB/s\n", rate(now() - t));
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sb, flag);
        t = now();
        CK(hipMemcpyAsync(pin0, d0, bytes, hipMemcpyDeviceToHost, sb));
        CK(hipStreamSynchronize(sb));
        std::printf("D2H pinned, 1 GiB, stream that just ran a kernel:    %6.1f GB/s\n", rate(now() - t));
        t = now();
        for (size_t o = 0; o < bytes; o += chunk) {
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sb, flag);
            CK(hipMemcpyAsync((char*)pin0 + o, (char*)d0 + o, chunk, hipMemcpyDeviceToHost, sb));
        }
        CK(hipStreamSynchronize(sb));
        std::printf("D2H pinned, 16 x (kernel + 64 MB) on one stream:     %6.1f GB/s\n", rate(now() - t));
        for (int grid : {32, 64, 128, 256, 1024}) {
            t = now();
            hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, sa, (u4*)pin0, (const u4*)d0, bytes / 16);
            CK(hipStreamSynchronize(sa));
            std::printf("D2H pinned by a copy kernel, %4d blocks:             %6.1f GB/s\n", grid, rate(now() - t));
        }
        t = now();
        CK(hipMemcpyAsync(d1, pin1, bytes, hipMemcpyHostToDevice, sb));
        CK(hipStreamSynchronize(sb));
        std::printf("H2D pinned, 1 GiB:                                   %6.1f GB/s\n", rate(now() - t));
        t = now();
        CK(hipMemcpyAsync(d1, pin1, bytes, hipMemcpyHostToDevice, sb));
        CK(hipMemcpyAsync(pin0, d0, bytes, hipMemcpyDeviceToHost, sa));
        CK(hipDeviceSynchronize());
        std::printf("H2D + D2H (two hipMemcpyAsync) at once:              %6.1f GB/s each way\n", rate(now() - t));
        t = now();
        CK(hipMemcpyAsync(d1, pin1, bytes, hipMemcpyHostToDevice, sb));
        hipLaunchKernelGGL(copy16, dim3(64), dim3(256), 0, sa, (u4*)pin0, (const u4*)d0, bytes / 16);
        CK(hipDeviceSynchronize());
        std::printf("H2D (hipMemcpyAsync) + D2H (copy kernel) at once:    %6.1f GB/s each way\n", rate(now() - t));
        t = now();
        for (size_t o = 0; o < bytes; o += chunk) {
            CK(hipMemcpyAsync((char*)d1 + o, (char*)pin1 + o, chunk, hipMemcpyHostToDevice, sb));
            hipLaunchKernelGGL(copy16, dim3(64), dim3(256), 0, sa, (u4*)((char*)pin0 + o), (const u4*)((char*)d0 + o), chunk / 16);
        }
        CK(hipDeviceSynchronize());
        std::printf("the same in 16 chunks of 64 MB:                      %6.1f GB/s each way\n", rate(now() - t));
        t = now();
        hipLaunchKernelGGL(copy16, dim3(64), dim3(256), 0, sb, (u4*)d1, (const u4*)pin1, bytes / 16);
        hipLaunchKernelGGL(copy16, dim3(64), dim3(256), 0, sa, (u4*)pin0, (const u4*)d0, bytes / 16);
        CK(hipDeviceSynchronize());
        std::printf("H2D + D2H, both by copy kernels, at once:            %6.1f GB/s each way\n", rate(now() - t));
        t = now();
        CK(hipMemcpyAsync(page, d0, bytes, hipMemcpyDeviceToHost, sc));
        CK(hipStreamSynchronize(sc));
        std::printf("D2H pageable, 1 GiB, copy-only stream:               %6.1f GB/s\n", rate(now() - t));
        t = now();
        for (size_t o = 0; o < bytes; o += chunk) CK(hipMemcpyAsync(page + o, (char*)d0 + o, chunk, hipMemcpyDeviceToHost, sc));
        CK(hipStreamSynchronize(sc));
        std::printf("D2H pageable, 16 x 64 MB, copy-only stream:          %6.1f GB/s\n", rate(now() - t));
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sb, flag);
        t = now();
        CK(hipMemcpyAsync(page, d0, bytes, hipMemcpyDeviceToHost, sb));
        CK(hipStreamSynchronize(sb));
        std::printf("D2H pageable, 1 GiB, stream that just ran a kernel:  %6.1f GB/s\n", rate(now() - t));
        t = now();
        CK(hipHostRegister(page, bytes, hipHostRegisterDefault));
        const double treg = now() - t;
        void* dp = nullptr;
        CK(hipHostGetDevicePointer(&dp, page, 0));
        t = now();
        hipLaunchKernelGGL(copy16, dim3(64), dim3(256), 0, sa, (u4*)dp, (const u4*)d0, bytes / 16);
        CK(hipStreamSynchronize(sa));
        const double tcp = now() - t;
        t = now();
        CK(hipHostUnregister(page));
        std::printf("pageable: hipHostRegister %.1f ms, copy kernel %.1f GB/s, hipHostUnregister %.1f ms\n", treg * 1e3, rate(tcp), (now() - t) * 1e3);
        std::printf("\n");
    }
    return 0;
}
