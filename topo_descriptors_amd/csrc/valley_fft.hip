// K7b: valley / ridge index for large kernels, by FFT.
//
// The direct kernel (valley.hip) costs taps x angles x planes per pixel and stages the reach of the
// largest rotated kernel in LDS: 6 s for 8192^2 at 67 px, 29 s at 101 px, nothing beyond ~119 px.
// The reference's own example runs the index at scales up to 100 km (scripts/
// compute_topo_descriptors.py:24-37, :66-82), i.e. kernels of hundreds of pixels, through
// scipy.signal.convolve's FFT branch.  This path does the same on the GPU with hipFFT (the one
// library call of the hot path: a 2-D real FFT is not worth hand-writing):
//
//   F = rfft2(Z)            Z = the block, normalised ((x - mean) / std) and zero-padded to P x Q,
//                           P >= rows + kmax - 1, Q >= nx + kmax - 1, both 5-smooth multiples of 32
//   per angle and plane:    K = the plane-sum kernel scattered into a P x Q image at minus its tap
//                           offsets (so the circular convolution is the correlation the direct
//                           kernel evaluates), R = irfft2(rfft2(K) F) / (P Q)
//   fold:                   maximum over the planes, then the same strict running maximum / angle
//                           update as the direct kernel; clip at 0 at the end.
//
// Cost: 2 FFTs per angle and plane whatever the kernel size.  Like the reference's FFT (and unlike
// the direct kernel) a NaN anywhere in the block reaches every output, and the float32 rounding
// depends on P x Q, so results of different row blocks agree to rounding, not bit for bit.
#include "common.hpp"

#include <hipfft/hipfft.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <utility>

namespace topo {

namespace {

constexpr int kThreads = 256;

#define TOPO_FFT(call)                                                                   \
    do {                                                                                 \
        hipfftResult r_ = (call);                                                        \
        if (r_ != HIPFFT_SUCCESS) {                                                      \
            set_error("%s failed with hipfftResult %d (%s:%d)", #call, (int)r_, __FILE__, __LINE__); \
            return TOPO_AMD_EHIP;                                                        \
        }                                                                                \
    } while (0)

// transform length >= n: a 5-smooth multiple of 32 (of 2 for short axes).  Lengths with a factor 7 or few
// factors of 2 are markedly slower in rocFFT: 8505 = 3^5 5 7 took 4.1 s where 8640 = 2^6 3^3 5 takes 2.0 s.
int good_size(int n) {
    const int step = n >= 256 ? 32 : 2;
    for (int m = (n + step - 1) / step * step;; m += step) {
        int r = m;
        for (int f : {2, 3, 5})
            while (r % f == 0) r /= f;
        if (r == 1) return m;
    }
}

struct Plans {
    hipfftHandle fwd = 0, inv = 0;
};
std::map<std::pair<int, int>, Plans>& plan_cache() {
    static std::map<std::pair<int, int>, Plans> cache;
    return cache;
}

int plans_for(int P, int Q, Plans* out) {
    auto& cache = plan_cache();
    auto it = cache.find({P, Q});
    if (it == cache.end()) {
        if (cache.size() >= 4) {  // a handful of shapes per process; drop the others' work buffers
            for (auto& kv : cache) {
                (void)hipfftDestroy(kv.second.fwd);
                (void)hipfftDestroy(kv.second.inv);
            }
            cache.clear();
        }
        Plans p;
        TOPO_FFT(hipfftPlan2d(&p.fwd, P, Q, HIPFFT_R2C));
        TOPO_FFT(hipfftPlan2d(&p.inv, P, Q, HIPFFT_C2R));
        TOPO_FFT(hipfftSetStream(p.fwd, ctx().compute));
        TOPO_FFT(hipfftSetStream(p.inv, ctx().compute));
        it = cache.emplace(std::make_pair(P, Q), p).first;
    }
    *out = it->second;
    return TOPO_AMD_OK;
}

__global__ __launch_bounds__(kThreads) void normalise_pad_kernel(const float* in, int in_rows, int nx, float mean,
                                                                 float stdev, float* z, int Q) {
    const int x = blockIdx.x * kThreads + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= Q) return;
    float v = 0.0f;
    if (y < in_rows && x < nx) v = (in[(size_t)y * nx + x] - mean) / stdev;  // as the direct kernel does
    z[(size_t)y * Q + x] = v;
}

// plane `q` of a kernel of side ks (taps: ks*ks float4) at minus its tap offsets, or zeros there again
__global__ __launch_bounds__(kThreads) void scatter_kernel(const float* taps, int ks, int q, bool clear, float* k,
                                                           int P, int Q) {
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= ks * ks) return;
    const int dy = idx / ks - ks / 2, dx = idx % ks - ks / 2;
    const int r = dy > 0 ? P - dy : -dy, c = dx > 0 ? Q - dx : -dx;
    k[(size_t)r * Q + c] = clear ? 0.0f : taps[(size_t)idx * 4 + q];
}

__global__ __launch_bounds__(kThreads) void spectrum_product_kernel(const float2* f, float2* g, size_t n, float scale) {
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float2 a = f[i], b = g[i];
    g[i] = make_float2((a.x * b.x - a.y * b.y) * scale, (a.x * b.y + a.y * b.x) * scale);
}

// response of plane q -> maximum over the planes -> running maximum over the angles
__global__ __launch_bounds__(kThreads) void fold_kernel(const float* resp, int Q, int row_off, int nx, float* tmp,
                                                        float* best, float* dir, float angle, int q, int n_planes) {
    const int x = blockIdx.x * kThreads + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= nx) return;
    const size_t o = (size_t)y * nx + x;
    const float v = resp[(size_t)(y + row_off) * Q + x];
    const float m = q == 0 ? v : fmaxf(tmp[o], v);
    if (q + 1 < n_planes) {
        tmp[o] = m;
    } else if (m > best[o]) {  // strict: the first angle that reaches the maximum keeps it (topo.py:438)
        best[o] = m;
        dir[o] = angle;
    }
}

__global__ __launch_bounds__(kThreads) void fill_kernel(float* p, size_t n, float v) {
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ __launch_bounds__(kThreads) void clip_kernel(float* p, size_t n) {
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < n) p[i] = fmaxf(p[i], 0.0f);  // clip(min=0), topo.py:446
}

dim3 grid1(size_t n) { return dim3((unsigned)((n + kThreads - 1) / kThreads)); }

}  // namespace

void valley_fft_release() {
    for (auto& kv : plan_cache()) {
        (void)hipfftDestroy(kv.second.fwd);
        (void)hipfftDestroy(kv.second.inv);
    }
    plan_cache().clear();
}

int launch_valley_ridge_fft(const Block& b, const float* taps, const int32_t* ksize, const float* angles,
                            int n_angles, int n_planes, int kmax, double mean, double stdev, float* norm_out,
                            float* dir_out) {
    Context& c = ctx();
    // only the rows the requested outputs reach (a seam strip of a shard does not transform the block)
    const int first = std::max(b.in_row0, b.out_row0 - kmax / 2);
    const int last = std::min(b.in_row0 + b.in_rows, b.out_row0 + b.out_rows + (kmax - 1 - kmax / 2));
    const float* in = b.in + (size_t)(first - b.in_row0) * b.nx;
    const int in_rows = last - first;
    const int P = good_size(in_rows + kmax - 1), Q = good_size(b.nx + kmax - 1);
    const size_t real_n = (size_t)P * Q, cplx_n = (size_t)P * (Q / 2 + 1), out_n = (size_t)b.out_rows * b.nx;
    Plans plans;
    TOPO_TRY(plans_for(P, Q, &plans));
    void *f = nullptr, *g = nullptr, *kimg = nullptr, *resp = nullptr, *d_taps = nullptr, *tmp = nullptr;
    TOPO_TRY(workspace(4, cplx_n * sizeof(float2), &f));
    TOPO_TRY(workspace(5, cplx_n * sizeof(float2), &g));
    TOPO_TRY(workspace(6, real_n * sizeof(float), &kimg));
    TOPO_TRY(workspace(7, real_n * sizeof(float), &resp));
    TOPO_TRY(workspace(1, (size_t)kmax * kmax * 4 * sizeof(float), &d_taps));
    TOPO_TRY(workspace(2, out_n * sizeof(float), &tmp));

    // Z goes through `resp` (free until the first inverse transform)
    hipLaunchKernelGGL(normalise_pad_kernel, dim3((Q + kThreads - 1) / kThreads, P), dim3(kThreads), 0, c.compute,
                       in, in_rows, b.nx, (float)mean, (float)stdev, (float*)resp, Q);
    TOPO_HIP(hipGetLastError());
    TOPO_FFT(hipfftExecR2C(plans.fwd, (hipfftReal*)resp, (hipfftComplex*)f));
    TOPO_HIP(hipMemsetAsync(kimg, 0, real_n * sizeof(float), c.compute));
    hipLaunchKernelGGL(fill_kernel, grid1(out_n), dim3(kThreads), 0, c.compute, norm_out, out_n, -INFINITY);
    hipLaunchKernelGGL(fill_kernel, grid1(out_n), dim3(kThreads), 0, c.compute, dir_out, out_n, 0.0f);
    TOPO_HIP(hipGetLastError());

    const float scale = (float)(1.0 / ((double)P * (double)Q));
    const int row_off = b.out_row0 - first;
    const dim3 out_grid((b.nx + kThreads - 1) / kThreads, b.out_rows);
    const float* src = taps;
    for (int a = 0; a < n_angles; ++a) {
        const int ks = ksize[a];
        const size_t ntap = (size_t)ks * ks;
        // the previous angle's kernels are done with d_taps: everything runs on one stream
        TOPO_HIP(hipMemcpyAsync(d_taps, src, ntap * 4 * sizeof(float), hipMemcpyHostToDevice, c.compute));
        src += ntap * 4;
        for (int q = 0; q < n_planes; ++q) {
            hipLaunchKernelGGL(scatter_kernel, grid1(ntap), dim3(kThreads), 0, c.compute, (const float*)d_taps, ks, q,
                               false, (float*)kimg, P, Q);
            TOPO_HIP(hipGetLastError());
            TOPO_FFT(hipfftExecR2C(plans.fwd, (hipfftReal*)kimg, (hipfftComplex*)g));
            hipLaunchKernelGGL(scatter_kernel, grid1(ntap), dim3(kThreads), 0, c.compute, (const float*)d_taps, ks, q,
                               true, (float*)kimg, P, Q);
            hipLaunchKernelGGL(spectrum_product_kernel, grid1(cplx_n), dim3(kThreads), 0, c.compute, (const float2*)f,
                               (float2*)g, cplx_n, scale);
            TOPO_HIP(hipGetLastError());
            TOPO_FFT(hipfftExecC2R(plans.inv, (hipfftComplex*)g, (hipfftReal*)resp));
            hipLaunchKernelGGL(fold_kernel, out_grid, dim3(kThreads), 0, c.compute, (const float*)resp, Q, row_off, b.nx,
                               (float*)tmp, norm_out, dir_out, angles[a], q, n_planes);
            TOPO_HIP(hipGetLastError());
        }
    }
    hipLaunchKernelGGL(clip_kernel, grid1(out_n), dim3(kThreads), 0, c.compute, norm_out, out_n);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

}  // namespace topo
