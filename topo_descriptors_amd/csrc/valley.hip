// K7: valley / ridge index.
//
// Replaces the angle loop of topo.valley_ridge (reference topo.py:431-447): for each of 180
// angles the reference convolves the normalised DEM, broadcast to one plane per flat fraction,
// with the stack of rotated V / U kernels in 3-D, takes the maximum over the planes, and keeps the
// largest value and the angle it came from.  Along the plane axis that 3-D "same" convolution
// only adds up neighbouring kernel planes (the DEM planes are identical), so the host hands over,
// per angle, n_planes 2-D kernels that are already those sums, flipped (so the device evaluates a
// correlation) and interleaved as one float4 per tap.
//
// Layout: a 64 x 16 output tile plus the reach of the largest rotated kernel is staged in LDS,
// already normalised ((x - mean) / std in float32, the two operations numpy performs) and zero
// outside the DEM (the reference zero-pads the normalised field).  Lanes run along x, each
// thread owns 4 rows; per tap the weights are one wave-uniform float4 (scalar load), and every
// row costs one conflict-free ds_read_b32 and n_planes FMAs.  The running maximum and its angle
// stay in registers, so the DEM is read once and the two outputs written once for all angles.
// Only 39-59 % of the taps of the rotated kernels are non-zero (the rotation enlarges the canvas and
// the reference masks the cells its spline contaminated), so the launcher hands the kernel a
// compressed list: weights plus the LDS offset of each non-zero tap.  Skipping exact zeros does not
// change a sum.
// Direct float32 evaluation on the vector ALU; cost ~ taps x angles x planes per pixel, so kernels of
// 64 px and more are handed to the FFT route (valley_fft.hip), as are those too large for the LDS tile.
// Round 6: kernels of up to 17 px (25 cells a side rotated) run as a dense product on the matrix pipe (valley_mfma.hip,
// 5 x as fast); this kernel then follows it in "repair" mode over the tiles that one flagged and rewrites exactly the pixels
// it marked (norm = -1: a non-finite sample in the kernel footprint) - and remains the evaluation of everything between 19
// and 63 px, and the reference of the matrix-pipe kernels' tests (TOPO_AMD_VALLEY_MFMA_MAX_KERNEL=0).
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace topo {

namespace {

constexpr int kThreads = 256;
constexpr int kTileW = 64;
constexpr int kTileH = 16;
constexpr int kRows = kTileH / (kThreads / 64);  // output rows per thread
// side of the largest rotated kernel from which the FFT path takes over (TOPO_AMD_VALLEY_FFT_MIN_KERNEL)
constexpr int kValleyFftFrom = 64;  // measured: direct 911 ms / FFT 1818 ms at 47 px, 6216 / 1818 at 95 px (8192^2)
typedef float tap4 __attribute__((ext_vector_type(4)));  // one tap: the weights of up to four planes
static_assert(kTileW == 64 && kValleyMfmaTileRows % kTileH == 0, "the repair pass maps its tiles onto valley_mfma.hip's");

struct VrArgs {
    const float* in;
    float* norm;
    float* dir;
    const tap4* taps;     // the non-zero taps of all angles back to back, one component per plane
    const int* tap_off;   // per tap: its offset in the LDS tile relative to the pixel's top-left reach corner
    const int* meta;      // per angle: number of non-zero taps, first tap
    const float* angles;  // value stored in dir for each angle
    int n_angles;
    int in_rows, in_row0, gny, nx;
    int out_row0, out_rows;
    int kmax, stride, rows_l, cols_l;
    float mean, stdev;
    const int* repair;  // not NULL: behind valley_mfma.hip - only the tiles it flagged, and in them only the pixels it marked norm = -1
    int repair_cols;
};

template <int NP>
__global__ __launch_bounds__(kThreads) void valley_ridge_kernel(VrArgs p) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ox0 = blockIdx.x * kTileW;
    const int oy0 = p.out_row0 + blockIdx.y * kTileH;
    const int reach = p.kmax / 2;  // "same" centring: a kernel of side K starts K / 2 before the pixel
    if (p.repair != nullptr && p.repair[(blockIdx.y / (kValleyMfmaTileRows / kTileH)) * p.repair_cols + blockIdx.x] == 0) return;

    for (int r = wave; r < p.rows_l; r += kThreads / 64) {
        const int gy = oy0 - reach + r;
        const int by = gy - p.in_row0;
        const bool row_ok = gy >= 0 && gy < p.gny && by >= 0 && by < p.in_rows;
        float* dst = L + r * p.stride;
        for (int k = lane; k < p.cols_l; k += 64) {
            const int gx = ox0 - reach + k;
            float v = 0.0f;
            if (row_ok && gx >= 0 && gx < p.nx) v = (p.in[(size_t)by * p.nx + gx] - p.mean) / p.stdev;
            dst[k] = v;
        }
    }
    __syncthreads();

    // wave-uniform tables through the constant address space: scalar loads
    typedef const __attribute__((address_space(4))) tap4* tap_ptr;
    typedef const __attribute__((address_space(4))) int* int_ptr;
    typedef const __attribute__((address_space(4))) float* flt_ptr;
    const int_ptr meta = (int_ptr)p.meta;
    const flt_ptr angles = (flt_ptr)p.angles;

    float best[kRows], best_angle[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        best[r] = -INFINITY;
        best_angle[r] = 0.0f;
    }
    for (int a = 0; a < p.n_angles; ++a) {
        const int ntap = meta[2 * a];
        const int first = meta[2 * a + 1];
        const tap_ptr w = (tap_ptr)p.taps + first;
        const int_ptr off = (int_ptr)p.tap_off + first;
        float acc[NP][kRows];
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int r = 0; r < kRows; ++r) acc[q][r] = 0.0f;
        const float* base = L + wave * p.stride + lane;
        for (int t = 0; t < ntap; ++t) {
            const tap4 wv = w[t];
            const float wq[4] = {wv[0], wv[1], wv[2], wv[3]};
            const float* at = base + off[t];
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const float z = at[(4 * r) * p.stride];
#pragma unroll
                for (int q = 0; q < NP; ++q) acc[q][r] = fmaf(wq[q], z, acc[q][r]);
            }
        }
        const float angle = angles[a];
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            float m = acc[0][r];
#pragma unroll
            for (int q = 1; q < NP; ++q) m = fmaxf(m, acc[q][r]);
            if (m > best[r]) {  // strict: the first angle that reaches the maximum keeps it (topo.py:438)
                best[r] = m;
                best_angle[r] = angle;
            }
        }
    }
    const int ox = ox0 + lane;
    if (ox >= p.nx) return;
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int oy = oy0 + wave + 4 * r;
        if (oy >= p.out_row0 + p.out_rows) continue;
        const size_t o = (size_t)(oy - p.out_row0) * p.nx + ox;
        if (p.repair != nullptr && p.norm[o] != -1.0f) continue;
        p.norm[o] = fmaxf(best[r], 0.0f);  // clip(min=0), topo.py:446
        p.dir[o] = best_angle[r];
    }
}

template <int NP>
int launch_np(const VrArgs& a, dim3 grid, size_t lds) {
    TOPO_HIP(hipFuncSetAttribute((const void*)valley_ridge_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    hipLaunchKernelGGL(valley_ridge_kernel<NP>, grid, dim3(kThreads), lds, ctx().compute, a);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

}  // namespace

// Sum and sum of squared deviations from a pivot (the first sample), float64, one partial pair
// per block; the host adds the partials.  Deviations from a pivot keep the variance of ~2000 m
// elevations with ~500 m spread free of cancellation.
namespace {
__global__ __launch_bounds__(kThreads) void moments_kernel(const float* in, size_t count, double pivot_value,
                                                           int pivot_is_first_sample, double* partial) {
    __shared__ double s1[kThreads], s2[kThreads];
    const double pivot = pivot_is_first_sample ? (double)in[0] : pivot_value;
    double a = 0.0, b = 0.0;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < count; i += (size_t)gridDim.x * kThreads) {
        const double d = (double)in[i] - pivot;
        a += d;
        b += d * d;
    }
    s1[threadIdx.x] = a;
    s2[threadIdx.x] = b;
    __syncthreads();
    for (int w = kThreads / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s1[threadIdx.x] += s1[threadIdx.x + w];
            s2[threadIdx.x] += s2[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = s1[0];
        partial[2 * blockIdx.x + 1] = s2[0];
    }
}
}  // namespace

// sum and sum of squares of (x - pivot) over count samples, float64, returned to the host
int launch_moments(const float* in, size_t count, double pivot, bool pivot_is_first_sample, double* sum,
                   double* sumsq) {
    Context& c = ctx();
    const int blocks = 1024;
    void* d_part = nullptr;
    TOPO_TRY(workspace(0, (size_t)blocks * 2 * sizeof(double), &d_part));
    hipLaunchKernelGGL(moments_kernel, dim3(blocks), dim3(kThreads), 0, c.compute, in, count, pivot,
                       pivot_is_first_sample ? 1 : 0, (double*)d_part);
    TOPO_HIP(hipGetLastError());
    std::vector<double> h((size_t)blocks * 2);
    TOPO_HIP(hipMemcpyAsync(h.data(), d_part, h.size() * sizeof(double), hipMemcpyDeviceToHost, c.compute));
    TOPO_HIP(hipStreamSynchronize(c.compute));
    double a = 0.0, b = 0.0;
    for (int i = 0; i < blocks; ++i) {
        a += h[2 * i];
        b += h[2 * i + 1];
    }
    *sum = a;
    *sumsq = b;
    return TOPO_AMD_OK;
}

int launch_mean_std(const float* in, size_t count, double* mean, double* stdev) {
    Context& c = ctx();
    float pivot = 0.0f;
    TOPO_HIP(hipMemcpyAsync(&pivot, in, sizeof(float), hipMemcpyDeviceToHost, c.compute));
    double a = 0.0, b = 0.0;
    TOPO_TRY(launch_moments(in, count, 0.0, true, &a, &b));
    const double n = (double)count;
    const double m = a / n;
    double var = b / n - m * m;
    if (var < 0.0) var = 0.0;
    *mean = (double)pivot + m;
    *stdev = std::sqrt(var);
    return TOPO_AMD_OK;
}

int valley_ridge_reach(const int32_t* ksize, int n_angles, int* above, int* below) {
    int kmax = 0;
    for (int a = 0; a < n_angles; ++a) kmax = ksize[a] > kmax ? ksize[a] : kmax;
    *above = kmax / 2;
    *below = kmax - 1 - kmax / 2;
    return kmax;
}

int launch_valley_ridge(const Block& b, const float* taps, const int32_t* ksize, const float* angles, int n_angles,
                        int n_planes, double mean, double stdev, float* norm_out, float* dir_out) {
    TOPO_REQUIRE(n_planes >= 1 && n_planes <= 4, "valley_ridge: %d kernel planes (1 to 4 are built)", n_planes);
    TOPO_REQUIRE(n_angles >= 1, "valley_ridge: no angles");
    TOPO_REQUIRE(stdev > 0.0 && stdev == stdev && mean == mean, "valley_ridge: mean %g / std %g of the DEM", mean, stdev);
    size_t dense = 0;
    for (int a = 0; a < n_angles; ++a) {
        TOPO_REQUIRE(ksize[a] >= 1, "valley_ridge: kernel side %d at angle index %d", ksize[a], a);
        dense += (size_t)ksize[a] * ksize[a];
        TOPO_REQUIRE(dense < ((size_t)1 << 30), "valley_ridge: kernel table too large");
    }
    int above = 0, below = 0;
    const int kmax = valley_ridge_reach(ksize, n_angles, &above, &below);
    VrArgs a{};
    a.in = b.in;
    a.norm = norm_out;
    a.dir = dir_out;
    a.n_angles = n_angles;
    a.in_rows = b.in_rows;
    a.in_row0 = b.in_row0;
    a.gny = b.gny;
    a.nx = b.nx;
    a.out_row0 = b.out_row0;
    a.out_rows = b.out_rows;
    a.kmax = kmax;
    a.rows_l = kTileH + kmax - 1;
    a.cols_l = kTileW + kmax - 1;
    a.stride = a.cols_l | 1;
    a.mean = (float)mean;
    a.stdev = (float)stdev;
    const size_t lds = (size_t)a.rows_l * a.stride * sizeof(float);
    // Which evaluation.  The matrix pipe (valley_mfma.hip) for everything it takes: rotated kernels of up to 120 cells a side when
    // the tables are point-symmetric, of up to 25 otherwise - this kernel then only visits the tiles in which that one met a
    // non-finite sample.  Else, large kernels by FFT (valley_fft.hip), whose cost does not depend on the kernel size: from 64
    // cells a side, or from TOPO_AMD_VALLEY_FFT_MIN_KERNEL - which, when set, also takes those kernels away from the matrix pipe
    // (the tests' way to the FFT route) - and whatever this kernel cannot stage.
    const char* e = std::getenv("TOPO_AMD_VALLEY_FFT_MIN_KERNEL");
    const bool fft_pinned = e && *e;
    const int fft_from = fft_pinned ? std::atoi(e) : kValleyFftFrom;
    const bool stageable = lds <= 160 * 1024;
    e = std::getenv("TOPO_AMD_VALLEY_MFMA_MAX_KERNEL");
    const int mfma_upto = e && *e ? std::min(std::atoi(e), kValleyStreamMaxKernel) : kValleyStreamMaxKernel;
    int done = 0;
    if (kmax <= mfma_upto && stageable && !(fft_pinned && kmax >= fft_from))
        TOPO_TRY(launch_valley_ridge_mfma(b, taps, ksize, angles, n_angles, n_planes, kmax, mean, stdev, norm_out, dir_out,
                                          &a.repair, &a.repair_cols, &done));
    if (!done) a.repair = nullptr;
    if (!done && (kmax >= fft_from || !stageable)) {
        note_valley_route(2);
        return launch_valley_ridge_fft(b, taps, ksize, angles, n_angles, n_planes, kmax, mean, stdev, norm_out, dir_out);
    }
    note_valley_route(done ? 1 + 4 + (done >= 2 ? 8 : 0) + (done == 3 ? 16 : 0) : 0);
    // compress: the non-zero taps of each angle, with their offset in the LDS tile (smaller kernels
    // sit centred inside the reach staged for the largest one)
    std::vector<int> meta((size_t)2 * n_angles);
    std::vector<float> wlist;
    std::vector<int> olist;
    wlist.reserve(dense * 2);
    olist.reserve(dense / 2);
    const float* src = taps;
    for (int ang = 0; ang < n_angles; ++ang) {
        const int ks = ksize[ang];
        const int shift = kmax / 2 - ks / 2;
        meta[2 * ang + 1] = (int)olist.size();
        for (int ky = 0; ky < ks; ++ky) {
            for (int kx = 0; kx < ks; ++kx, src += 4) {
                bool any = false;
                for (int q = 0; q < n_planes; ++q) any = any || src[q] != 0.0f;
                if (!any) continue;
                wlist.insert(wlist.end(), src, src + 4);
                olist.push_back((ky + shift) * a.stride + kx + shift);
            }
        }
        meta[2 * ang] = (int)olist.size() - meta[2 * ang + 1];
    }
    if (olist.empty()) {  // keep the tables non-empty for the uploads
        wlist.assign(4, 0.0f);
        olist.assign(1, 0);
    }
    void *d_taps = nullptr, *d_off = nullptr, *d_meta = nullptr, *d_angles = nullptr;
    TOPO_TRY(upload_table(0, wlist.data(), wlist.size() * sizeof(float), &d_taps));
    TOPO_TRY(upload_table(3, olist.data(), olist.size() * sizeof(int), &d_off));
    TOPO_TRY(upload_table(1, meta.data(), meta.size() * sizeof(int), &d_meta));
    TOPO_TRY(upload_table(2, angles, (size_t)n_angles * sizeof(float), &d_angles));
    a.taps = (const tap4*)d_taps;
    a.tap_off = (const int*)d_off;
    a.meta = (const int*)d_meta;
    a.angles = (const float*)d_angles;
    dim3 grid((b.nx + kTileW - 1) / kTileW, (b.out_rows + kTileH - 1) / kTileH);
    switch (n_planes) {
        case 1: return launch_np<1>(a, grid, lds);
        case 2: return launch_np<2>(a, grid, lds);
        case 3: return launch_np<3>(a, grid, lds);
        default: return launch_np<4>(a, grid, lds);
    }
}

}  // namespace topo
