// The order of operations of csrc/capi.hip's run_pipelined on its own: H2D chunks on one stream, a small kernel per chunk
// on a second, D2H chunks on a third - issued (A) all from one host thread, (B) with the downloads from a second thread
// that waits on a condition variable, as run_pipelined did first.  Page-locked and pageable host memory.
//   hipcc --offload-arch=gfx950 -O2 -pthread -o pipe_paths.bin pipe_paths.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(float* out, const float* in, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i] + 1.0f;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    const int nch = 16;
    const size_t chunk = bytes / nch;
    float *d_in = nullptr, *d_out = nullptr;
    hipMalloc((void**)&d_in, bytes);
    hipMalloc((void**)&d_out, bytes);
    void *pin_in = nullptr, *pin_out = nullptr;
    hipHostMalloc(&pin_in, bytes, hipHostMallocDefault);
    hipHostMalloc(&pin_out, bytes, hipHostMallocDefault);
    std::memset(pin_in, 1, bytes);
    std::memset(pin_out, 2, bytes);
    char* page_in = (char*)std::malloc(bytes);
    char* page_out = (char*)std::malloc(bytes);
    std::memset(page_in, 1, bytes);
    std::memset(page_out, 2, bytes);
    hipStream_t up, comp, down;
    hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&comp, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&down, hipStreamNonBlocking);
    std::vector<hipEvent_t> eu(nch), ec(nch);
    for (auto& e : eu) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (auto& e : ec) hipEventCreateWithFlags(&e, hipEventDisableTiming);

    auto serial = [&](const char* in, char* out) {
        hipMemcpyAsync(d_in, in, bytes, hipMemcpyHostToDevice, comp);
        hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, comp, d_out, d_in, bytes / 4);
        hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, comp);
        hipStreamSynchronize(comp);
    };
    auto one_thread = [&](const char* in, char* out) {
        for (int k = 0; k < nch; ++k) {
            const size_t o = k * chunk;
            hipMemcpyAsync((char*)d_in + o, in + o, chunk, hipMemcpyHostToDevice, up);
            hipEventRecord(eu[k], up);
            hipStreamWaitEvent(comp, eu[k], 0);
            hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, comp, (float*)((char*)d_out + o), (const float*)((char*)d_in + o), chunk / 4);
            hipEventRecord(ec[k], comp);
            hipStreamWaitEvent(down, ec[k], 0);
            hipMemcpyAsync(out + o, (char*)d_out + o, chunk, hipMemcpyDeviceToHost, down);
        }
        hipStreamSynchronize(down);
        hipStreamSynchronize(comp);
    };
    auto two_threads = [&](const char* in, char* out) {
        std::mutex mu;
        std::condition_variable cv;
        int ready = 0;
        std::thread dl([&] {
            hipSetDevice(0);
            for (int j = 0; j < nch; ++j) {
                {
                    std::unique_lock<std::mutex> lock(mu);
                    cv.wait(lock, [&] { return ready > j; });
                }
                const size_t o = j * chunk;
                hipStreamWaitEvent(down, ec[j], 0);
                hipMemcpyAsync(out + o, (char*)d_out + o, chunk, hipMemcpyDeviceToHost, down);
            }
            hipStreamSynchronize(down);
        });
        for (int k = 0; k < nch; ++k) {
            const size_t o = k * chunk;
            hipMemcpyAsync((char*)d_in + o, in + o, chunk, hipMemcpyHostToDevice, up);
            hipEventRecord(eu[k], up);
            hipStreamWaitEvent(comp, eu[k], 0);
            hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, comp, (float*)((char*)d_out + o), (const float*)((char*)d_in + o), chunk / 4);
            hipEventRecord(ec[k], comp);
            {
                std::lock_guard<std::mutex> lock(mu);
                ready = k + 1;
            }
            cv.notify_all();
        }
        dl.join();
        hipStreamSynchronize(comp);
    };
    // downloads by a thread that synchronises on the event itself and then copies with the blocking call
    auto two_threads_sync = [&](const char* in, char* out) {
        std::mutex mu;
        std::condition_variable cv;
        int ready = 0;
        std::thread dl([&] {
            hipSetDevice(0);
            for (int j = 0; j < nch; ++j) {
                {
                    std::unique_lock<std::mutex> lock(mu);
                    cv.wait(lock, [&] { return ready > j; });
                }
                const size_t o = j * chunk;
                hipEventSynchronize(ec[j]);
                hipMemcpy(out + o, (char*)d_out + o, chunk, hipMemcpyDeviceToHost);
            }
        });
        for (int k = 0; k < nch; ++k) {
            const size_t o = k * chunk;
            hipMemcpyAsync((char*)d_in + o, in + o, chunk, hipMemcpyHostToDevice, up);
            hipEventRecord(eu[k], up);
            hipStreamWaitEvent(comp, eu[k], 0);
            hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, comp, (float*)((char*)d_out + o), (const float*)((char*)d_in + o), chunk / 4);
            hipEventRecord(ec[k], comp);
            {
                std::lock_guard<std::mutex> lock(mu);
                ready = k + 1;
            }
            cv.notify_all();
        }
        dl.join();
        hipStreamSynchronize(comp);
    };
    struct Case {
        const char* name;
        const char* in;
        char* out;
    } cases[] = {{"page-locked", (const char*)pin_in, (char*)pin_out}, {"pageable", page_in, page_out}};
    for (int rep = 0; rep < 2; ++rep)
        for (const Case& c : cases) {
            double t = now();
            serial(c.in, c.out);
            std::printf("%-12s serial on one stream:                 %6.2f ms\n", c.name, (now() - t) * 1e3);
            t = now();
            one_thread(c.in, c.out);
            std::printf("%-12s 16 chunks, three streams, one thread: %6.2f ms\n", c.name, (now() - t) * 1e3);
            t = now();
            two_threads(c.in, c.out);
            std::printf("%-12s ... downloads from a second thread:   %6.2f ms\n", c.name, (now() - t) * 1e3);
            t = now();
            two_threads_sync(c.in, c.out);
            std::printf("%-12s ... second thread, blocking hipMemcpy: %6.2f ms\n", c.name, (now() - t) * 1e3);
        }
    return 0;
}
