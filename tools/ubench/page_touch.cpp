// First-touch cost of a fresh result buffer on the host (what the host-buffer entry points'
// callers pay for the output array): malloc, mmap, mmap + MADV_HUGEPAGE, mmap + MAP_POPULATE,
// then a device-to-host copy into each.   build: hipcc -O2 -o page_touch.bin page_touch.cpp
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    void* dev = nullptr;
    if (hipMalloc(&dev, bytes) != hipSuccess) return 1;
    hipMemset(dev, 0, bytes);
    hipDeviceSynchronize();
    auto d2h = [&](const char* what, void* p) {
        double t = now();
        hipMemcpy(p, dev, bytes, hipMemcpyDeviceToHost);
        double dt = now() - t;
        std::printf("%-44s D2H into it: %7.1f ms (%.1f GB/s)\n", what, dt * 1e3, bytes / dt / 1e9);
        t = now();
        hipMemcpy(p, dev, bytes, hipMemcpyDeviceToHost);
        dt = now() - t;
        std::printf("%-44s   again:     %7.1f ms (%.1f GB/s)\n", "", dt * 1e3, bytes / dt / 1e9);
    };
    {
        void* p = std::malloc(bytes);
        d2h("malloc, untouched", p);
        std::free(p);
    }
    {
        double t = now();
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        int rc = madvise(p, bytes, MADV_HUGEPAGE);
        std::printf("mmap + MADV_HUGEPAGE rc %d: %.2f ms\n", rc, (now() - t) * 1e3);
        d2h("mmap + MADV_HUGEPAGE, untouched", p);
        munmap(p, bytes);
    }
    {
        double t = now();
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
        std::printf("mmap + MAP_POPULATE: %.1f ms\n", (now() - t) * 1e3);
        d2h("mmap + MAP_POPULATE", p);
        munmap(p, bytes);
    }
    {
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(p, bytes, MADV_HUGEPAGE);
        double t = now();
        std::memset(p, 0, bytes);
        std::printf("memset of mmap + MADV_HUGEPAGE: %.1f ms\n", (now() - t) * 1e3);
        munmap(p, bytes);
    }
    {
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        double t = now();
        std::memset(p, 0, bytes);
        std::printf("memset of plain mmap: %.1f ms\n", (now() - t) * 1e3);
        munmap(p, bytes);
    }
    return 0;
}
