// What host-side options the host-buffer entry points have: cost of pinning (hipHostMalloc,
// hipHostRegister) against the copy rates of pageable and pinned memory, both directions, and
// both directions at once.   build: hipcc -O2 -o pin_rate pin_rate.cpp    run: ./pin_rate [GiB=1]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            std::printf("%s: %s\n", #x, hipGetErrorString(e));                     \
            return 1;                                                              \
        }                                                                          \
    } while (0)

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)((argc > 1 ? std::atof(argv[1]) : 1.0) * (1 << 30));
    void *d0 = nullptr, *d1 = nullptr;
    CK(hipMalloc(&d0, bytes));
    CK(hipMalloc(&d1, bytes));
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0));
    CK(hipStreamCreate(&s1));

    char* pageable = (char*)std::malloc(bytes);
    double t = now();
    std::memset(pageable, 1, bytes);
    std::printf("first touch of %.2f GiB malloc: %.1f ms\n", bytes / 1073741824.0, (now() - t) * 1e3);
    char* pageable2 = (char*)std::malloc(bytes);
    std::memset(pageable2, 1, bytes);
    t = now();
    std::memcpy(pageable2, pageable, bytes);
    std::printf("memcpy host to host, one thread: %.1f GB/s\n", bytes / (now() - t) / 1e9);

    for (int rep = 0; rep < 2; ++rep) {
        t = now();
        CK(hipMemcpy(d0, pageable, bytes, hipMemcpyHostToDevice));
        std::printf("pageable H2D: %.1f GB/s\n", bytes / (now() - t) / 1e9);
        t = now();
        CK(hipMemcpy(pageable2, d0, bytes, hipMemcpyDeviceToHost));
        std::printf("pageable D2H: %.1f GB/s\n", bytes / (now() - t) / 1e9);
    }

    void* pinned = nullptr;
    t = now();
    CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
    std::printf("hipHostMalloc: %.1f ms (%.1f GB/s)\n", (now() - t) * 1e3, bytes / (now() - t) / 1e9);
    void* pinned2 = nullptr;
    CK(hipHostMalloc(&pinned2, bytes, hipHostMallocDefault));
    t = now();
    std::memcpy(pinned, pageable, bytes);
    std::printf("memcpy pageable to pinned (first touch): %.1f GB/s\n", bytes / (now() - t) / 1e9);
    for (int rep = 0; rep < 2; ++rep) {
        t = now();
        CK(hipMemcpy(d0, pinned, bytes, hipMemcpyHostToDevice));
        std::printf("pinned H2D: %.1f GB/s\n", bytes / (now() - t) / 1e9);
        t = now();
        CK(hipMemcpy(pinned2, d1, bytes, hipMemcpyDeviceToHost));
        std::printf("pinned D2H: %.1f GB/s\n", bytes / (now() - t) / 1e9);
        t = now();
        CK(hipMemcpyAsync(d0, pinned, bytes, hipMemcpyHostToDevice, s0));
        CK(hipMemcpyAsync(pinned2, d1, bytes, hipMemcpyDeviceToHost, s1));
        CK(hipDeviceSynchronize());
        std::printf("pinned H2D + D2H at once: %.1f GB/s each way\n", bytes / (now() - t) / 1e9);
    }
    t = now();
    CK(hipHostFree(pinned2));
    std::printf("hipHostFree: %.1f ms\n", (now() - t) * 1e3);

    t = now();
    CK(hipHostRegister(pageable, bytes, hipHostRegisterDefault));
    std::printf("hipHostRegister (touched pages): %.1f ms (%.1f GB/s)\n", (now() - t) * 1e3, bytes / (now() - t) / 1e9);
    t = now();
    CK(hipMemcpy(d0, pageable, bytes, hipMemcpyHostToDevice));
    std::printf("registered H2D: %.1f GB/s\n", bytes / (now() - t) / 1e9);
    t = now();
    CK(hipHostUnregister(pageable));
    std::printf("hipHostUnregister: %.1f ms\n", (now() - t) * 1e3);
    char* fresh = (char*)std::malloc(bytes);
    t = now();
    CK(hipHostRegister(fresh, bytes, hipHostRegisterDefault));
    std::printf("hipHostRegister (untouched pages): %.1f ms\n", (now() - t) * 1e3);
    t = now();
    CK(hipMemcpy(fresh, d1, bytes, hipMemcpyDeviceToHost));
    std::printf("registered D2H: %.1f GB/s\n", bytes / (now() - t) / 1e9);
    CK(hipHostUnregister(fresh));
    return 0;
}
