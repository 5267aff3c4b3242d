// K6: Sx - maximum elevation angle along rays in an azimuth sector.
//
// Replaces the numba loop _sx_rolling (topo.py:928-953).  The host hands over the ray pixels
// as an offset table (dj, di, distance); duplicates are removed and NaN distances (left by
// radius_min, topo.py:845) dropped here because max() ignores both.  atan is monotone, so
// the kernel maximises the tangent (dem[p] - dem[j,i] - height) / dist[p] and takes ONE atan
// per pixel instead of one per ray point.  A frame of `window` pixels stays 0 like the
// reference's zeros_like() output (topo.py:939-941).
//
// Layout: a tile of the DEM plus the bounding box of the offsets is staged in LDS; lanes run
// along x, the offset table is wave-uniform (scalar loads), so every step is one conflict-
// free ds_read_b32 + v_sub + v_mul + v_max per lane.
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <utility>

namespace topo {

namespace {

constexpr int kThreads = 256;
constexpr int kTileW = 64;
constexpr int kTileH = 32;  // 8 output rows per thread

struct SxArgs {
    const float* in;
    float* out;
    const int* lds_off;     // per unique offset: (dj - dj_min) * stride + (di - di_min)
    const float* inv_dist;  // 1 / distance in metres
    int n_off;
    int in_rows, in_row0, gny, nx;
    int out_row0, out_rows;
    int window;
    int dj_min, di_min, rows_l, cols_l, stride;
    float height;
};

__global__ __launch_bounds__(kThreads) void sx_kernel(SxArgs p) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ox0 = blockIdx.x * kTileW;
    const int oy0 = p.out_row0 + blockIdx.y * kTileH;

    // stage tile + offset bounding box; pixels outside the DEM are never used by interior
    // outputs (the zero frame is exactly as wide as the reach of the rays)
    for (int r = wave; r < p.rows_l; r += kThreads / 64) {
        const int gy = oy0 + p.dj_min + r;
        const int by = gy - p.in_row0;
        const bool row_ok = gy >= 0 && gy < p.gny && by >= 0 && by < p.in_rows;
        float* dst = L + r * p.stride;
        for (int k = lane; k < p.cols_l; k += 64) {
            const int gx = ox0 + p.di_min + k;
            dst[k] = (row_ok && gx >= 0 && gx < p.nx) ? p.in[(size_t)by * p.nx + gx] : 0.0f;
        }
    }
    __syncthreads();

    const int ox = ox0 + lane;
    constexpr int NOUT = kTileH / (kThreads / 64);
    float best[NOUT], centre[NOUT];
    const int self = -p.dj_min * p.stride - p.di_min + lane;
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
        best[k] = -INFINITY;
        centre[k] = L[self + (wave + 4 * k) * p.stride] + p.height;
    }
    for (int n = 0; n < p.n_off; ++n) {
        const int off = p.lds_off[n] + lane;  // wave-uniform table entries
        const float inv = p.inv_dist[n];
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            const float z = L[off + (wave + 4 * k) * p.stride] - centre[k];
            best[k] = fmaxf(best[k], z * inv);  // fmaxf drops NaN operands like nanmax
        }
    }
    if (ox >= p.nx) return;
    const float rad2deg = 57.29577951308232f;
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
        const int oy = oy0 + wave + 4 * k;
        if (oy >= p.out_row0 + p.out_rows) continue;
        const bool inside = oy >= p.window && oy < p.gny - p.window && ox >= p.window &&
                            ox < p.nx - p.window;
        float v = 0.0f;
        if (inside) v = best[k] == -INFINITY ? NAN : atanf(best[k]) * rad2deg;
        p.out[(size_t)(oy - p.out_row0) * p.nx + ox] = v;
    }
}

// Search windows too large for an LDS tile: the same scan straight from global memory (the ray
// pixels of neighbouring lanes are neighbours, so every read is a coalesced row segment that L2
// serves after the first touch).
struct SxGlobalArgs {
    const float* in;
    float* out;
    const int* dj;
    const int* di;
    const float* inv_dist;
    int n_off;
    int in_row0, gny, nx, out_row0, out_rows, window;
    float height;
};

__global__ __launch_bounds__(kThreads) void sx_global_kernel(SxGlobalArgs p) {
    const int ox = blockIdx.x * kThreads + threadIdx.x;
    const int oy = p.out_row0 + blockIdx.y;
    if (ox >= p.nx) return;
    const size_t o = (size_t)(oy - p.out_row0) * p.nx + ox;
    const bool inside = oy >= p.window && oy < p.gny - p.window && ox >= p.window && ox < p.nx - p.window;
    if (!inside) {
        p.out[o] = 0.0f;
        return;
    }
    const float centre = p.in[(size_t)(oy - p.in_row0) * p.nx + ox] + p.height;
    float best = -INFINITY;
    for (int n = 0; n < p.n_off; ++n) {
        const float v = p.in[(size_t)(oy + p.dj[n] - p.in_row0) * p.nx + ox + p.di[n]];
        best = fmaxf(best, (v - centre) * p.inv_dist[n]);
    }
    p.out[o] = best == -INFINITY ? NAN : atanf(best) * 57.29577951308232f;
}

__global__ __launch_bounds__(kThreads) void fill_kernel(float* out, size_t n, float value) {
    size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    const size_t step = (size_t)gridDim.x * kThreads;
    for (; i < n; i += step) out[i] = value;
}

// integer hash -> [0, 1)
__device__ __forceinline__ float hash01(uint32_t a, uint32_t b, uint32_t seed) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ seed * 0xC2B2AE3Du;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 12;
    h *= 0x297A2D39u;
    h ^= h >> 15;
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(kThreads) void synth_kernel(float* out, int rows, int row0, int nx,
                                                         uint32_t seed, bool integer_valued) {
    const int x = blockIdx.x * kThreads + threadIdx.x;
    const int r = blockIdx.y;
    if (x >= nx || r >= rows) return;
    const float j = (float)(row0 + r), i = (float)x;
    float z = 1900.0f + 520.0f * __sinf(j / 211.0f) * __cosf(i / 173.0f) +
              310.0f * __sinf((j + 2.0f * i) / 97.0f) + 120.0f * __cosf((3.0f * j - i) / 41.0f) +
              40.0f * __sinf(j / 9.0f) * __sinf(i / 7.0f);
    // ~5 m noise: sum of four uniforms, variance 4/12 -> scale to sigma 5
    float u = hash01(row0 + r, x, seed) + hash01(row0 + r, x, seed + 1) +
              hash01(row0 + r, x, seed + 2) + hash01(row0 + r, x, seed + 3) - 2.0f;
    z += u * 8.660254f;
    out[(size_t)r * nx + x] = integer_valued ? rintf(z) : z;
}

}  // namespace

int launch_synth(float* out, int rows, int row0, int nx, uint32_t seed, bool integer_valued) {
    Context& c = ctx();
    dim3 grid((nx + kThreads - 1) / kThreads, rows);
    hipLaunchKernelGGL(synth_kernel, grid, dim3(kThreads), 0, c.compute, out, rows, row0, nx, seed,
                       integer_valued);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

int launch_sx(const Block& b, const int32_t* dj, const int32_t* di, const double* dist, int n_off,
              int window, double height, float* out) {
    Context& c = ctx();
    TOPO_REQUIRE(window >= 0, "sx: negative window %d", window);
    // unique (dj, di) with a finite, usable distance
    std::vector<std::pair<std::pair<int, int>, double>> pts;
    pts.reserve(n_off);
    for (int n = 0; n < n_off; ++n) {
        if (std::isnan(dist[n])) continue;
        TOPO_REQUIRE(std::abs(dj[n]) <= window && std::abs(di[n]) <= window,
                     "sx: offset (%d, %d) reaches beyond the zero frame of width %d", dj[n],
                     di[n], window);
        pts.push_back({{dj[n], di[n]}, dist[n]});
    }
    std::sort(pts.begin(), pts.end());
    pts.erase(std::unique(pts.begin(), pts.end(),
                          [](const auto& a, const auto& b2) { return a.first == b2.first; }),
              pts.end());
    const size_t total = (size_t)b.out_rows * b.nx;
    if (pts.empty()) {
        // nanmax over nothing: NaN inside the frame (numpy warns and returns NaN)
        int blocks = (int)std::min<size_t>((total + kThreads - 1) / kThreads, 4096);
        hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(kThreads), 0, c.compute, out, total, 0.0f);
        TOPO_HIP(hipGetLastError());
        set_error("sx: no usable ray pixel (all distances NaN)");
        return TOPO_AMD_EINVAL;
    }
    SxArgs a;
    int dj_min = 0, dj_max = 0, di_min = 0, di_max = 0;  // the target pixel itself is staged too
    for (auto& q : pts) {
        dj_min = std::min(dj_min, q.first.first);
        dj_max = std::max(dj_max, q.first.first);
        di_min = std::min(di_min, q.first.second);
        di_max = std::max(di_max, q.first.second);
    }
    a.dj_min = dj_min;
    a.di_min = di_min;
    a.rows_l = kTileH + dj_max - dj_min;
    a.cols_l = kTileW + di_max - di_min;
    a.stride = a.cols_l | 1;
    const size_t lds = (size_t)a.rows_l * a.stride * sizeof(float);
    if (lds > 160 * 1024) {
        std::vector<int> vdj(pts.size()), vdi(pts.size());
        std::vector<float> vinv(pts.size());
        for (size_t n = 0; n < pts.size(); ++n) {
            vdj[n] = pts[n].first.first;
            vdi[n] = pts[n].first.second;
            vinv[n] = (float)(1.0 / pts[n].second);
        }
        void *d_dj = nullptr, *d_di = nullptr, *d_inv = nullptr;
        TOPO_TRY(upload_table(0, vdj.data(), vdj.size() * sizeof(int), &d_dj));
        TOPO_TRY(upload_table(1, vdi.data(), vdi.size() * sizeof(int), &d_di));
        TOPO_TRY(upload_table(2, vinv.data(), vinv.size() * sizeof(float), &d_inv));
        SxGlobalArgs ga{b.in, out, (const int*)d_dj, (const int*)d_di, (const float*)d_inv, (int)pts.size(),
                        b.in_row0, b.gny, b.nx, b.out_row0, b.out_rows, window, (float)height};
        dim3 ggrid((b.nx + kThreads - 1) / kThreads, b.out_rows);
        hipLaunchKernelGGL(sx_global_kernel, ggrid, dim3(kThreads), 0, c.compute, ga);
        TOPO_HIP(hipGetLastError());
        return TOPO_AMD_OK;
    }
    std::vector<int> off(pts.size());
    std::vector<float> inv(pts.size());
    for (size_t n = 0; n < pts.size(); ++n) {
        off[n] = (pts[n].first.first - dj_min) * a.stride + (pts[n].first.second - di_min);
        inv[n] = (float)(1.0 / pts[n].second);
    }
    void *d_off = nullptr, *d_inv = nullptr;
    TOPO_TRY(upload_table(0, off.data(), off.size() * sizeof(int), &d_off));
    TOPO_TRY(upload_table(1, inv.data(), inv.size() * sizeof(float), &d_inv));
    a.in = b.in;
    a.out = out;
    a.lds_off = (const int*)d_off;
    a.inv_dist = (const float*)d_inv;
    a.n_off = (int)pts.size();
    a.in_rows = b.in_rows;
    a.in_row0 = b.in_row0;
    a.gny = b.gny;
    a.nx = b.nx;
    a.out_row0 = b.out_row0;
    a.out_rows = b.out_rows;
    a.window = window;
    a.height = (float)height;
    dim3 grid((b.nx + kTileW - 1) / kTileW, (b.out_rows + kTileH - 1) / kTileH);
    TOPO_HIP(hipFuncSetAttribute((const void*)sx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    hipLaunchKernelGGL(sx_kernel, grid, dim3(kThreads), lds, c.compute, a);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

}  // namespace topo
