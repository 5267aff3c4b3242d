// Does the copy rate depend on WHICH stream a copy is issued on?  (The pipelined host-buffer path saw D2H at 29 GB/s on its
// download stream - the fifth stream the library creates - where a fresh process copies at 56 GB/s on its first stream.)
// Creates 8 non-blocking streams (the second one with the highest priority, like the library's RCCL stream) and times a
// 256 MB page-locked D2H and H2D on each, in creation order.
//   hipcc --offload-arch=gfx950 -O2 -o stream_engines.bin stream_engines.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t bytes = (size_t)256 << 20;
    void *d = nullptr, *h = nullptr;
    hipMalloc(&d, bytes);
    hipHostMalloc(&h, bytes, hipHostMallocDefault);
    std::memset(h, 1, bytes);
    hipMemset(d, 2, bytes);
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStream_t s[8];
    for (int k = 0; k < 8; ++k) {
        if (k == 1) hipStreamCreateWithPriority(&s[k], hipStreamNonBlocking, greatest);
        else hipStreamCreateWithFlags(&s[k], hipStreamNonBlocking);
    }
    for (int rep = 0; rep < 3; ++rep) {
        for (int k = 0; k < 8; ++k) {
            double t = now();
            hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s[k]);
            hipStreamSynchronize(s[k]);
            const double d2h = bytes / (now() - t) / 1e9;
            t = now();
            hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s[k]);
            hipStreamSynchronize(s[k]);
            const double h2d = bytes / (now() - t) / 1e9;
            std::printf("rep %d stream %d: D2H %5.1f GB/s   H2D %5.1f GB/s\n", rep, k, d2h, h2d);
        }
    }
    return 0;
}
