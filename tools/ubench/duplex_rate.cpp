// Can a host-buffer call overlap its upload with its download?  Two host threads copy pageable
// (touched) memory in opposite directions at the same time, in chunks, each on its own stream.
// build: hipcc -O2 -pthread -o duplex_rate.bin duplex_rate.cpp
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const size_t bytes = (size_t)2 << 30, chunk = (size_t)256 << 20;
    void *d0 = nullptr, *d1 = nullptr;
    hipMalloc(&d0, bytes);
    hipMalloc(&d1, bytes);
    char* up = (char*)std::malloc(bytes);
    char* down = (char*)std::malloc(bytes);
    std::memset(up, 1, bytes);
    std::memset(down, 1, bytes);
    auto h2d = [&] {
        hipSetDevice(0);
        hipStream_t s;
        hipStreamCreate(&s);
        for (size_t o = 0; o < bytes; o += chunk) {
            hipMemcpyAsync((char*)d0 + o, up + o, chunk, hipMemcpyHostToDevice, s);
            hipStreamSynchronize(s);
        }
    };
    auto d2h = [&] {
        hipSetDevice(0);
        hipStream_t s;
        hipStreamCreate(&s);
        for (size_t o = 0; o < bytes; o += chunk) {
            hipMemcpyAsync(down + o, (char*)d1 + o, chunk, hipMemcpyDeviceToHost, s);
            hipStreamSynchronize(s);
        }
    };
    for (int rep = 0; rep < 2; ++rep) {
        double t = now();
        h2d();
        std::printf("upload alone:   %.1f GB/s\n", bytes / (now() - t) / 1e9);
        t = now();
        d2h();
        std::printf("download alone: %.1f GB/s\n", bytes / (now() - t) / 1e9);
        t = now();
        std::thread a(h2d), b(d2h);
        a.join();
        b.join();
        std::printf("both at once:   %.1f GB/s each way (%.1f ms for 2 x %.0f GiB)\n", bytes / (now() - t) / 1e9,
                    (now() - t) * 1e3, bytes / 1073741824.0);
    }
    return 0;
}
