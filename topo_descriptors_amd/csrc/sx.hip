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
#include <cstdlib>
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

// ---- several azimuth sectors in one pass ------------------------------------------
// The reference computes one azimuth per call (topo.py:715-772, looped by its users); sectors of
// neighbouring azimuths overlap (arc 10 degrees, usual step 5), so most ray pixels belong to two of
// them.  The launcher sorts the unique ray pixels of up to kMaxAz sectors into classes of equal
// membership; the kernel stages the tile once, scans every class once with the loop of sx_kernel
// and folds the class maximum into the sectors that contain it.  max() does not care about the
// order, so each plane has the bits of the single-azimuth kernel.
constexpr int kMaxAz = 8;

struct SxMultiArgs {
    const float* in;
    float* out[kMaxAz];
    int window[kMaxAz];
    const int* lds_off;     // per unique ray pixel, sorted by class
    const float* inv_dist;
    const int* cls_first;   // n_cls + 1 entries
    const int* cls_mask;    // bit a: sector a contains the class
    int n_cls, n_az;
    int in_rows, in_row0, gny, nx;
    int out_row0, out_rows;
    int dj_min, di_min, rows_l, cols_l, stride;
    float height;
};

template <int NA>
__global__ __launch_bounds__(kThreads) void sx_multi_kernel(SxMultiArgs p) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ox0 = blockIdx.x * kTileW;
    const int oy0 = p.out_row0 + blockIdx.y * kTileH;

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
    float best[NA][NOUT], centre[NOUT];
    const int self = -p.dj_min * p.stride - p.di_min + lane;
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
        centre[k] = L[self + (wave + 4 * k) * p.stride] + p.height;
#pragma unroll
        for (int a = 0; a < NA; ++a) best[a][k] = -INFINITY;
    }
    for (int c = 0; c < p.n_cls; ++c) {
        float top[NOUT];
#pragma unroll
        for (int k = 0; k < NOUT; ++k) top[k] = -INFINITY;
        const int n1 = p.cls_first[c + 1];
        for (int n = p.cls_first[c]; n < n1; ++n) {
            const int off = p.lds_off[n] + lane;  // wave-uniform table entries
            const float inv = p.inv_dist[n];
#pragma unroll
            for (int k = 0; k < NOUT; ++k) {
                const float z = L[off + (wave + 4 * k) * p.stride] - centre[k];
                top[k] = fmaxf(top[k], z * inv);  // fmaxf drops NaN operands like nanmax
            }
        }
        const int mask = p.cls_mask[c];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            if (mask >> a & 1) {
#pragma unroll
                for (int k = 0; k < NOUT; ++k) best[a][k] = fmaxf(best[a][k], top[k]);
            }
        }
    }
    if (ox >= p.nx) return;
    const float rad2deg = 57.29577951308232f;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        if (a >= p.n_az) break;
        const int w = p.window[a];
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            const int oy = oy0 + wave + 4 * k;
            if (oy >= p.out_row0 + p.out_rows) continue;
            const bool inside = oy >= w && oy < p.gny - w && ox >= w && ox < p.nx - w;
            float v = 0.0f;
            if (inside) v = best[a][k] == -INFINITY ? NAN : atanf(best[a][k]) * rad2deg;
            p.out[a][(size_t)(oy - p.out_row0) * p.nx + ox] = v;
        }
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
    TOPO_TRY(check_grid_rows(rows, "synth_dem"));
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
        return TOPO_AMD_EEMPTY;
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

namespace {

struct SectorPoints {
    std::vector<std::pair<std::pair<int, int>, double>> pts;  // unique usable (dj, di), distance
    int dj_min = 0, dj_max = 0, di_min = 0, di_max = 0;       // box, the target pixel included
};

size_t sx_tile_bytes(int dj_min, int dj_max, int di_min, int di_max) {
    const int rows_l = kTileH + dj_max - dj_min, cols_l = kTileW + di_max - di_min;
    return (size_t)rows_l * (cols_l | 1) * sizeof(float);
}

// one launch of sx_multi_kernel for sectors [a0, a1)
int launch_sx_group(const Block& b, const std::vector<SectorPoints>& sec, int a0, int a1,
                    const int32_t* window, double height, float* const* outs) {
    Context& c = ctx();
    std::vector<std::pair<std::pair<int, int>, std::pair<int, double>>> all;  // (dj, di) -> (mask, dist)
    SxMultiArgs a;
    a.dj_min = a.di_min = 0;
    int dj_max = 0, di_max = 0;
    for (int k = a0; k < a1; ++k) {
        for (auto& q : sec[k].pts) all.push_back({q.first, {1 << (k - a0), q.second}});
        a.dj_min = std::min(a.dj_min, sec[k].dj_min);
        a.di_min = std::min(a.di_min, sec[k].di_min);
        dj_max = std::max(dj_max, sec[k].dj_max);
        di_max = std::max(di_max, sec[k].di_max);
    }
    std::sort(all.begin(), all.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
    size_t w = 0;  // merge equal offsets: union of the masks (the distance depends on the offset only)
    for (size_t n = 0; n < all.size(); ++n) {
        if (w > 0 && all[w - 1].first == all[n].first) {
            all[w - 1].second.first |= all[n].second.first;
        } else {
            all[w++] = all[n];
        }
    }
    all.resize(w);
    std::stable_sort(all.begin(), all.end(),
                     [](const auto& x, const auto& y) { return x.second.first < y.second.first; });
    a.rows_l = kTileH + dj_max - a.dj_min;
    a.cols_l = kTileW + di_max - a.di_min;
    a.stride = a.cols_l | 1;
    const size_t lds = (size_t)a.rows_l * a.stride * sizeof(float);
    std::vector<int> off(all.size()), first, mask;
    std::vector<float> inv(all.size());
    for (size_t n = 0; n < all.size(); ++n) {
        off[n] = (all[n].first.first - a.dj_min) * a.stride + (all[n].first.second - a.di_min);
        inv[n] = (float)(1.0 / all[n].second.second);
        if (n == 0 || all[n].second.first != all[n - 1].second.first) {
            first.push_back((int)n);
            mask.push_back(all[n].second.first);
        }
    }
    first.push_back((int)all.size());
    void *d_off = nullptr, *d_inv = nullptr, *d_first = nullptr, *d_mask = nullptr;
    TOPO_TRY(upload_table(0, off.data(), off.size() * sizeof(int), &d_off));
    TOPO_TRY(upload_table(1, inv.data(), inv.size() * sizeof(float), &d_inv));
    TOPO_TRY(upload_table(2, first.data(), first.size() * sizeof(int), &d_first));
    TOPO_TRY(upload_table(3, mask.data(), mask.size() * sizeof(int), &d_mask));
    a.in = b.in;
    for (int k = 0; k < kMaxAz; ++k) {
        a.out[k] = k < a1 - a0 ? outs[a0 + k] : nullptr;
        a.window[k] = k < a1 - a0 ? window[a0 + k] : 0;
    }
    a.lds_off = (const int*)d_off;
    a.inv_dist = (const float*)d_inv;
    a.cls_first = (const int*)d_first;
    a.cls_mask = (const int*)d_mask;
    a.n_cls = (int)mask.size();
    a.n_az = a1 - a0;
    a.in_rows = b.in_rows;
    a.in_row0 = b.in_row0;
    a.gny = b.gny;
    a.nx = b.nx;
    a.out_row0 = b.out_row0;
    a.out_rows = b.out_rows;
    a.height = (float)height;
    dim3 grid((b.nx + kTileW - 1) / kTileW, (b.out_rows + kTileH - 1) / kTileH);
    auto go = [&](auto kernel) -> int {
        TOPO_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, grid, dim3(kThreads), lds, c.compute, a);
        TOPO_HIP(hipGetLastError());
        return TOPO_AMD_OK;
    };
    if (a.n_az <= 2) return go(sx_multi_kernel<2>);
    if (a.n_az <= 4) return go(sx_multi_kernel<4>);
    return go(sx_multi_kernel<kMaxAz>);
}

}  // namespace

int launch_sx_multi(const Block& b, int n_az, const int32_t* first, const int32_t* dj, const int32_t* di,
                    const double* dist, const int32_t* window, double height, float* const* outs) {
    // at most this much LDS per block, so that three blocks still share a CU
    static const size_t kGroupLds = [] {
        const char* e = std::getenv("TOPO_AMD_SX_GROUP_LDS_KIB");
        const int kib = e && *e ? std::atoi(e) : 48;
        return (size_t)std::min(std::max(kib, 1), 160) * 1024;
    }();
    std::vector<SectorPoints> sec(n_az);
    for (int k = 0; k < n_az; ++k) {
        TOPO_REQUIRE(window[k] >= 0 && first[k + 1] >= first[k], "sx_multi: bad sector %d", k);
        SectorPoints& s = sec[k];
        for (int n = first[k]; n < first[k + 1]; ++n) {
            if (std::isnan(dist[n])) continue;
            TOPO_REQUIRE(std::abs(dj[n]) <= window[k] && std::abs(di[n]) <= window[k],
                         "sx_multi: offset (%d, %d) reaches beyond the zero frame of width %d", dj[n], di[n],
                         window[k]);
            s.pts.push_back({{dj[n], di[n]}, dist[n]});
        }
        std::sort(s.pts.begin(), s.pts.end());
        s.pts.erase(std::unique(s.pts.begin(), s.pts.end(),
                                [](const auto& x, const auto& y) { return x.first == y.first; }),
                    s.pts.end());
        for (auto& q : s.pts) {
            s.dj_min = std::min(s.dj_min, q.first.first);
            s.dj_max = std::max(s.dj_max, q.first.first);
            s.di_min = std::min(s.di_min, q.first.second);
            s.di_max = std::max(s.di_max, q.first.second);
        }
    }
    int rc = TOPO_AMD_OK;
    for (int a0 = 0; a0 < n_az;) {
        // neighbouring sectors while their common tile stays small; a sector on its own (or one
        // without a usable ray pixel) takes the single-azimuth path, whatever its size
        int a1 = a0 + 1;
        int dj_min = sec[a0].dj_min, dj_max = sec[a0].dj_max, di_min = sec[a0].di_min, di_max = sec[a0].di_max;
        while (a1 < n_az && a1 - a0 < kMaxAz && !sec[a0].pts.empty() && !sec[a1].pts.empty()) {
            const int j0 = std::min(dj_min, sec[a1].dj_min), j1 = std::max(dj_max, sec[a1].dj_max);
            const int i0 = std::min(di_min, sec[a1].di_min), i1 = std::max(di_max, sec[a1].di_max);
            if (sx_tile_bytes(j0, j1, i0, i1) > kGroupLds) break;
            dj_min = j0, dj_max = j1, di_min = i0, di_max = i1;
            ++a1;
        }
        if (a1 - a0 == 1) {
            const int n0 = first[a0];
            const int r = launch_sx(b, dj + n0, di + n0, dist + n0, first[a0 + 1] - n0, window[a0], height, outs[a0]);
            if (r != TOPO_AMD_OK) rc = r;  // an empty sector: reported at the end, the others still run
        } else {
            TOPO_TRY(launch_sx_group(b, sec, a0, a1, window, height, outs));
        }
        a0 = a1;
    }
    return rc;
}

}  // namespace topo
