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
#include "gate.hpp"
#include "atan.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

namespace topo {

namespace {

constexpr int kThreads = 256;

// Table entries: chains of ray pixels that are neighbours along the chain axis (rows for sectors that point
// north / south, columns for those that point east / west).
struct SxChain8 {
    int off;       // LDS index of the first ray pixel of the chain, relative to the target pixel's tile origin
    int pad[7];
    float inv[8];  // 1 / distance in metres of pixel m of the chain; NaN past the end of a shorter chain
};
struct SxChain4 {
    int off;
    float inv[4];
    int pad[3];
};
struct SxChain2 {
    int off;
    float inv[2];
    int pad;
};
static_assert(sizeof(SxChain8) == 64 && sizeof(SxChain4) == 32 && sizeof(SxChain2) == 16, "scalar-load friendly entries");

struct SxArgs {
    const float* in;
    float* out;
    const SxChain8* tab8;
    const SxChain4* tab4;
    const SxChain2* tab2;
    int n8, n4, n2;
    // pairs of chains with the same weights (pad[0] / pad of the entry: the LDS index of the second chain)
    const SxChain8* tab8p;
    const SxChain4* tab4p;
    const SxChain2* tab2p;
    int n8p, n4p, n2p;
    int in_rows, in_row0, gny, nx;
    int out_row0, out_rows;
    int window;
    int dj_min, di_min, rows_l, cols_l;
    float height;
    // row shards (common.hpp, GhostGate): tile rows 0 ... n_top - 1 read ghost rows above the shard, the last n_bot
    // ones ghost rows below it; they are dispatched last and wait at the gate.  tiles_y: tile rows of the launch
    // (gridDim.y, except in the clean-up launch, whose grid holds the n_bot + n_top ghost tile rows only).
    int tiles_y, n_top, n_bot, cleanup;
    Gate gate;
};

constexpr int kSxTile = 64;                     // output tile: 64 x 64 pixels
constexpr int kSxOwn = kSxTile / (kThreads / 64);  // consecutive pixels per lane along the chain axis

// A lane owns kSxOwn consecutive pixels along the chain axis (ALONG_X false: 16 rows of one column, lanes along x;
// true: 16 columns of one row, lanes along y).  The ray pixels of a sector are dense in (dj, di): along the axis
// the sector points in they come in long runs of neighbours, and the sample a lane needs for (own pixel k, ray
// pixel m + 1) is the one it needs for (own pixel k + 1, ray pixel m).  So the table holds chains of 8 (and, for
// what is left of a run, 2) neighbouring ray pixels: 8 + kSxOwn - 1 LDS reads feed 8 kSxOwn comparisons (0.18
// reads per comparison; one read each made the kernel LDS-bound: ds_read_b32 moves 128 B/clk/CU, 25 ms of reads
// at radius 2000 m).  What is left is the arithmetic, 3 issue slots per pixel and ray point: v_sub, v_mul, and one
// v_max3 per two points (v_max_f32 / v_max3_f32 issue at half rate on gfx950, profiles/r02_valu_mix_rate.txt).
// The LDS row stride is a template parameter so that the samples of a chain are immediates on one address
// register.  A chain shorter than its table entry ends in NaN weights: the products are NaN and max3 drops them
// like nanmax; what such a chain reads past the tile is spare LDS.
// NW waves: 4 (tile 64 x 64) or, lanes along x only, 8 (tile 64 columns x 128 rows: a third less halo per output and
// two blocks of 8 waves per CU where the 64-row tile has three of 4).
// DIAG = +1 / -1 (round 3; lanes along x only): the chains run along a DIAGONAL - the sample for (own pixel k, ray
// pixel m + 1) is again the one for (own pixel k + 1, ray pixel m) when both the own pixels of a lane and the ray
// pixels of a chain step (1 row, DIAG columns) - so the runs of a sector that points north-east ... are as long as
// those of one that points north.  The code is the same with the LDS step STRIDE + DIAG; the 16 pixels of a lane
// lean over 15 columns (every wave's slab of 16 rows starts upright again, so the staged tile is 15 columns wider
// and the tile grid one tile longer), and for a fixed k the lanes still store 64 consecutive columns of one row.
// GATED (row shards): the tile rows that read ghost rows come last in the dispatch order and wait at the ghost-row gate
// (a second instantiation: the few scalars it keeps alive cost the ordinary kernel 4 ... 19 % at radius 500 ... 2000 m).
template <int STRIDE, bool ALONG_X, int NW, int DIAG = 0, bool GATED = false>
__global__ __launch_bounds__(NW * 64) void sx_kernel(SxArgs p) {
    static_assert(NW == 4 || !ALONG_X, "8 waves: lanes along x only");
    static_assert(DIAG == 0 || !ALONG_X, "diagonal chains: lanes along x only");
    constexpr int SPAN = kSxOwn * NW;  // tile extent along the chain axis
    constexpr int XS = DIAG < 0 ? kSxOwn - 1 : 0;  // own pixels lean left: the tile is staged that many columns further left
    extern __shared__ __attribute__((aligned(16))) float L[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ox0 = ((int)blockIdx.x - (DIAG > 0 ? 1 : 0)) * (ALONG_X ? SPAN : kSxTile);
    // tile row: the rows that read ghost rows come last in the dispatch order (top ones after the bottom ones)
    int ty = (int)blockIdx.y;
    if (GATED) {
        ty += p.n_top;
        ty = ty >= p.tiles_y ? ty - p.tiles_y : ty;
        if (p.cleanup) ty = (int)blockIdx.y < p.n_bot ? p.tiles_y - p.n_bot + (int)blockIdx.y : (int)blockIdx.y - p.n_bot;
    }
    const int oy0 = p.out_row0 + ty * (ALONG_X ? kSxTile : SPAN);
    if (GATED && p.gate.word != nullptr && (ty < p.n_top || ty >= p.tiles_y - p.n_bot)) {
        const int rank = ty >= p.tiles_y - p.n_bot ? ty - (p.tiles_y - p.n_bot) : p.n_bot + ty;
        const unsigned slot = (unsigned)rank * gridDim.x + blockIdx.x;
        if (p.cleanup) {
            if (p.gate.skipped[slot] == 0) return;
            __syncthreads();
            if (threadIdx.x == 0) p.gate.skipped[slot] = 0;
        } else if (!gate_wait(p.gate, slot) && p.gate.errors == nullptr) {
            return;  // left to the clean-up launch behind the exchange's event
        }
    }

    // stage tile + offset bounding box; pixels outside the DEM are never used by interior
    // outputs (the zero frame is exactly as wide as the reach of the rays).  Four rows per wave in flight.
    // (a tile whose staged rows and columns all lie inside the block - nearly every tile - needs no test per sample)
    const int gx_first = ox0 + p.di_min - XS, gy_first = oy0 + p.dj_min;
    const bool tile_inside = gx_first >= 0 && gx_first + p.cols_l <= p.nx && gy_first >= max(0, p.in_row0) &&
                             gy_first + p.rows_l <= min(p.gny, p.in_row0 + p.in_rows);
    if (tile_inside) {
        const float* src = p.in + (size_t)(gy_first - p.in_row0) * p.nx + gx_first;
        for (int r0 = 0; r0 < p.rows_l; r0 += 4 * NW) {
            for (int k0 = 0; k0 < p.cols_l; k0 += 64) {
                const int k = k0 + lane;
                if (k < p.cols_l) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = min(r0 + wave + NW * u, p.rows_l - 1);
                        v[u] = src[(size_t)r * p.nx + k];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = r0 + wave + NW * u;
                        if (r < p.rows_l) L[r * STRIDE + k] = v[u];
                    }
                }
            }
        }
    } else {
        for (int r0 = 0; r0 < p.rows_l; r0 += 4 * NW) {
            for (int k0 = 0; k0 < p.cols_l; k0 += 64) {
                const int k = k0 + lane, gx = gx_first + k;
                const bool col_ok = k < p.cols_l && gx >= 0 && gx < p.nx;
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = r0 + wave + NW * u;
                    const int gy = gy_first + r, by = gy - p.in_row0;
                    const bool ok = col_ok && r < p.rows_l && gy >= 0 && gy < p.gny && by >= 0 && by < p.in_rows;
                    v[u] = ok ? p.in[(size_t)by * p.nx + gx] : 0.0f;
                }
                if (k < p.cols_l) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = r0 + wave + NW * u;
                        if (r < p.rows_l) L[r * STRIDE + k] = v[u];
                    }
                }
            }
        }
    }
    __syncthreads();

    constexpr int S = ALONG_X ? 1 : STRIDE + DIAG;  // LDS step along the chain axis
    float best[kSxOwn], centre[kSxOwn];
    // own pixel k of this lane at LDS index Lw + k S (+ the offset of a ray pixel)
    const float* Lw = ALONG_X ? L + lane * STRIDE + wave * kSxOwn : L + wave * kSxOwn * STRIDE + lane + XS;
    const int self = -p.dj_min * STRIDE - p.di_min;
#pragma unroll
    for (int k = 0; k < kSxOwn; ++k) {
        best[k] = -INFINITY;
        centre[k] = Lw[self + k * S] + p.height;
    }
    for (int c = 0; c < p.n8; ++c) {
        const SxChain8 e = p.tab8[c];  // wave-uniform: scalar loads
        const float* q = Lw + e.off;
        float w[kSxOwn + 7];
#pragma unroll
        for (int j = 0; j < kSxOwn + 7; ++j) w[j] = q[j * S];
#pragma unroll
        for (int m = 0; m < 8; m += 2) {
#pragma unroll
            for (int k = 0; k < kSxOwn; ++k) {
                const float z0 = (w[m + k] - centre[k]) * e.inv[m];
                const float z1 = (w[m + 1 + k] - centre[k]) * e.inv[m + 1];
                // max3 drops (quiet) NaN operands like nanmax; spelled out because the compiler turns
                // fmaxf(best, fmaxf(z0, z1)) into v_max(z0, z1) pairs folded by a later v_max3: 1.5 instead of 1
                // half-rate instruction per two products
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(best[k]) : "v"(z0), "v"(z1));
            }
        }
    }
    // chains of 4 (round 3): the runs of 3 ... 6 neighbours that diagonal sectors are made of went in twos (or in a
    // padded eight) before: 460 comparisons for the 358 ray pixels of azimuth 45 at radius 2000 m
    for (int c = 0; c < p.n4; ++c) {
        const SxChain4 e = p.tab4[c];
        const float* q = Lw + e.off;
        float w[kSxOwn + 3];
#pragma unroll
        for (int j = 0; j < kSxOwn + 3; ++j) w[j] = q[j * S];
#pragma unroll
        for (int m = 0; m < 4; m += 2) {
#pragma unroll
            for (int k = 0; k < kSxOwn; ++k) {
                const float z0 = (w[m + k] - centre[k]) * e.inv[m];
                const float z1 = (w[m + 1 + k] - centre[k]) * e.inv[m + 1];
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(best[k]) : "v"(z0), "v"(z1));
            }
        }
    }
    for (int c = 0; c < p.n2; c += 4) {  // four entries per scalar load (the table is padded to a multiple of 4)
        struct Quad {
            SxChain2 e[4];
        };
        const Quad g = *reinterpret_cast<const Quad*>(p.tab2 + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* q = Lw + g.e[u].off;
            float w[kSxOwn + 1];
#pragma unroll
            for (int j = 0; j < kSxOwn + 1; ++j) w[j] = q[j * S];
#pragma unroll
            for (int k = 0; k < kSxOwn; ++k) {
                const float z0 = (w[k] - centre[k]) * g.e[u].inv[0];
                const float z1 = (w[1 + k] - centre[k]) * g.e[u].inv[1];
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(best[k]) : "v"(z0), "v"(z1));
            }
        }
    }
    // Pairs (round 3).  A sector that points along an axis is symmetric about it: the ray pixels (dj, di) and (dj, -di)
    // are equally far, so their chains carry the same weights, and since (w - c) inv grows with w the larger of the two
    // SAMPLES decides - one v_max per sample (8 + 15 of them) instead of a second set of 8 x 16 products and their
    // maxima.  The launcher pairs any two chains of one length whose weights are equal bit for bit.  NaN: v_max
    // returns the other operand, as the two separate products would have been dropped one by one.
    auto larger = [](float a, float b) {
        float m;
        asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
        return m;
    };
    for (int c = 0; c < p.n8p; ++c) {
        const SxChain8 e = p.tab8p[c];
        const float* qa = Lw + e.off;
        const float* qb = Lw + e.pad[0];
        float w[kSxOwn + 7];
#pragma unroll
        for (int j = 0; j < kSxOwn + 7; ++j) w[j] = larger(qa[j * S], qb[j * S]);
#pragma unroll
        for (int m = 0; m < 8; m += 2) {
#pragma unroll
            for (int k = 0; k < kSxOwn; ++k) {
                const float z0 = (w[m + k] - centre[k]) * e.inv[m];
                const float z1 = (w[m + 1 + k] - centre[k]) * e.inv[m + 1];
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(best[k]) : "v"(z0), "v"(z1));
            }
        }
    }
    for (int c = 0; c < p.n4p; ++c) {
        const SxChain4 e = p.tab4p[c];
        const float* qa = Lw + e.off;
        const float* qb = Lw + e.pad[0];
        float w[kSxOwn + 3];
#pragma unroll
        for (int j = 0; j < kSxOwn + 3; ++j) w[j] = larger(qa[j * S], qb[j * S]);
#pragma unroll
        for (int m = 0; m < 4; m += 2) {
#pragma unroll
            for (int k = 0; k < kSxOwn; ++k) {
                const float z0 = (w[m + k] - centre[k]) * e.inv[m];
                const float z1 = (w[m + 1 + k] - centre[k]) * e.inv[m + 1];
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(best[k]) : "v"(z0), "v"(z1));
            }
        }
    }
    for (int c = 0; c < p.n2p; ++c) {
        const SxChain2 e = p.tab2p[c];
        const float* qa = Lw + e.off;
        const float* qb = Lw + e.pad;
        float w[kSxOwn + 1];
#pragma unroll
        for (int j = 0; j < kSxOwn + 1; ++j) w[j] = larger(qa[j * S], qb[j * S]);
#pragma unroll
        for (int k = 0; k < kSxOwn; ++k) {
            const float z0 = (w[k] - centre[k]) * e.inv[0];
            const float z1 = (w[1 + k] - centre[k]) * e.inv[1];
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(best[k]) : "v"(z0), "v"(z1));
        }
    }
    const float rad2deg = 57.29577951308232f;
    if (ALONG_X) {
        // lanes run along y here: the tile goes through LDS once more so that the stores are row segments
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kSxOwn; ++k) L[lane * (kSxTile + 1) + wave * kSxOwn + k] = best[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kSxOwn; ++k) best[k] = L[(wave * kSxOwn + k) * (kSxTile + 1) + lane];
    }
#pragma unroll
    for (int k = 0; k < kSxOwn; ++k) {
        const int ox = ox0 + lane + DIAG * k;
        if (ox < 0 || ox >= p.nx) continue;
        const int oy = oy0 + wave * kSxOwn + k;
        if (oy >= p.out_row0 + p.out_rows) continue;
        const bool inside = oy >= p.window && oy < p.gny - p.window && ox >= p.window &&
                            ox < p.nx - p.window;
        float v = 0.0f;
        if (inside) v = best[k] == -INFINITY ? NAN : atan_signed(best[k]) * rad2deg;
        p.out[(size_t)(oy - p.out_row0) * p.nx + ox] = v;
    }
}

// the strides the kernel is built for: the smallest one that holds the tile's columns is used
constexpr int kSxStrides[] = {67, 71, 75, 81, 89, 97, 105, 113, 129, 145, 161, 177, 193, 209, 225, 257};
constexpr int kSxStrideCount = sizeof(kSxStrides) / sizeof(kSxStrides[0]);

// blocks_per_cu != nullptr: no launch, only how many blocks of this kernel a CU holds
template <int I = 0>
int launch_sx_stride(int stride, bool along_x, int waves, dim3 grid, size_t lds, hipStream_t stream, const SxArgs& a, int diag = 0,
                     int* blocks_per_cu = nullptr) {
    if constexpr (I < kSxStrideCount) {
        if (stride == kSxStrides[I]) {
            auto go = [&](auto kernel) -> int {
                TOPO_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                if (blocks_per_cu) {
                    TOPO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, (const void*)kernel, waves * 64, lds));
                    return TOPO_AMD_OK;
                }
                hipLaunchKernelGGL(kernel, grid, dim3(waves * 64), lds, stream, a);
                TOPO_HIP(hipGetLastError());
                return TOPO_AMD_OK;
            };
            if (a.gate.word != nullptr) {
                if (along_x) return go(sx_kernel<kSxStrides[I], true, 4, 0, true>);
                if (diag > 0) return go(sx_kernel<kSxStrides[I], false, 4, 1, true>);
                if (diag < 0) return go(sx_kernel<kSxStrides[I], false, 4, -1, true>);
                return waves == 8 ? go(sx_kernel<kSxStrides[I], false, 8, 0, true>) : go(sx_kernel<kSxStrides[I], false, 4, 0, true>);
            }
            if (along_x) return go(sx_kernel<kSxStrides[I], true, 4>);
            if (diag > 0) return go(sx_kernel<kSxStrides[I], false, 4, 1>);
            if (diag < 0) return go(sx_kernel<kSxStrides[I], false, 4, -1>);
            return waves == 8 ? go(sx_kernel<kSxStrides[I], false, 8>) : go(sx_kernel<kSxStrides[I], false, 4>);
        }
        return launch_sx_stride<I + 1>(stride, along_x, waves, grid, lds, stream, a, diag, blocks_per_cu);
    } else {
        set_error("sx: no kernel for LDS stride %d", stride);
        return TOPO_AMD_EUNSUP;
    }
}

int sx_stride_for(int cols_l) {
    for (int s : kSxStrides)
        if (s >= cols_l) return s;
    return 0;
}

// Cut the unique ray pixels into chains along one axis: runs of neighbours in eights, what is left of a run (or
// a run shorter than 7) in fours and twos, so that at most one comparison per run is padding.  `along_x`: chains run along di (same dj), else along dj (same di).
typedef std::vector<std::pair<std::pair<int, int>, double>> SxPoints;
// `mode`: 0 chains along dj (same di), 1 along di (same dj), 2 / 3 along the diagonals (dj + 1, di + 1) / (dj + 1, di - 1).
void sx_chains(SxPoints pts, int mode, int stride, int dj_min, int di_min, std::vector<SxChain8>* t8,
               std::vector<SxChain4>* t4, std::vector<SxChain2>* t2) {
    auto line = [&](const SxPoints::value_type& q) {
        const int dj = q.first.first, di = q.first.second;
        return mode == 0 ? di : mode == 1 ? dj : mode == 2 ? di - dj : di + dj;
    };
    auto pos = [&](const SxPoints::value_type& q) { return mode == 1 ? q.first.second : q.first.first; };
    std::sort(pts.begin(), pts.end(), [&](const auto& x, const auto& y) {
        return std::make_pair(line(x), pos(x)) < std::make_pair(line(y), pos(y));
    });
    t8->clear();
    t4->clear();
    t2->clear();
    for (size_t n = 0; n < pts.size();) {
        size_t run = 1;
        while (n + run < pts.size() && line(pts[n + run]) == line(pts[n]) && pos(pts[n + run]) == pos(pts[n]) + (int)run) ++run;
        for (size_t done = 0; done < run;) {
            const size_t left = run - done, first = n + done;
            const int off = (pts[first].first.first - dj_min) * stride + (pts[first].first.second - di_min);
            if (left >= 7) {
                SxChain8 e{};
                e.off = off;
                const size_t m = std::min<size_t>(8, left);
                for (size_t z = 0; z < 8; ++z) e.inv[z] = z < m ? (float)(1.0 / pts[first + z].second) : std::nanf("");
                t8->push_back(e);
                done += m;
            } else if (left >= 3) {  // 3 ... 6: a four (and a two for what is left of 5 or 6)
                SxChain4 e{};
                e.off = off;
                const size_t m = std::min<size_t>(4, left);
                for (size_t z = 0; z < 4; ++z) e.inv[z] = z < m ? (float)(1.0 / pts[first + z].second) : std::nanf("");
                t4->push_back(e);
                done += m;
            } else {
                SxChain2 e{};
                e.off = off;
                e.inv[0] = (float)(1.0 / pts[first].second);
                e.inv[1] = left >= 2 ? (float)(1.0 / pts[first + 1].second) : std::nanf("");
                t2->push_back(e);
                done += std::min<size_t>(2, left);
            }
        }
        n += run;
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
    const SxChain8* tab8;   // chains sorted by class
    const SxChain4* tab4;
    const SxChain2* tab2;   // (every class padded to a multiple of 4 entries)
    const int* cls_first8;  // n_cls + 1 entries each
    const int* cls_first4;
    const int* cls_first2;
    const int* cls_mask;    // bit a: sector a contains the class
    int n_cls, n_az;
    int in_rows, in_row0, gny, nx;
    int out_row0, out_rows;
    int dj_min, di_min, rows_l, cols_l;
    float height;
};

constexpr int kSxMultiOwn = 8;                                  // pixels per lane along the chain axis
constexpr int kSxMultiSpan = kSxMultiOwn * (kThreads / 64);     // tile extent along the chain axis (32)

// Tile kSxMultiSpan x 64: ALONG_X false: 32 rows x 64 columns, lanes along x; true: 64 rows x 32 columns, lanes
// along y.  The scan of a class is the chain scan of sx_kernel into a class maximum.
template <int STRIDE, bool ALONG_X, int NA>
__global__ __launch_bounds__(kThreads) void sx_multi_kernel(SxMultiArgs p) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    constexpr int OWN = kSxMultiOwn;
    constexpr int TW = ALONG_X ? kSxMultiSpan : 64, TH = ALONG_X ? 64 : kSxMultiSpan;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ox0 = blockIdx.x * TW;
    const int oy0 = p.out_row0 + blockIdx.y * TH;

    // (as in sx_kernel: no test per sample for a tile that lies inside the block)
    const int gx_first = ox0 + p.di_min, gy_first = oy0 + p.dj_min;
    const bool tile_inside = gx_first >= 0 && gx_first + p.cols_l <= p.nx && gy_first >= max(0, p.in_row0) &&
                             gy_first + p.rows_l <= min(p.gny, p.in_row0 + p.in_rows);
    if (tile_inside) {
        const float* src = p.in + (size_t)(gy_first - p.in_row0) * p.nx + gx_first;
        for (int r0 = 0; r0 < p.rows_l; r0 += 16) {
            for (int k0 = 0; k0 < p.cols_l; k0 += 64) {
                const int k = k0 + lane;
                if (k < p.cols_l) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = src[(size_t)min(r0 + wave + 4 * u, p.rows_l - 1) * p.nx + k];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = r0 + wave + 4 * u;
                        if (r < p.rows_l) L[r * STRIDE + k] = v[u];
                    }
                }
            }
        }
    } else {
        for (int r0 = 0; r0 < p.rows_l; r0 += 16) {
            for (int k0 = 0; k0 < p.cols_l; k0 += 64) {
                const int k = k0 + lane, gx = gx_first + k;
                const bool col_ok = k < p.cols_l && gx >= 0 && gx < p.nx;
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = r0 + wave + 4 * u;
                    const int gy = gy_first + r, by = gy - p.in_row0;
                    const bool ok = col_ok && r < p.rows_l && gy >= 0 && gy < p.gny && by >= 0 && by < p.in_rows;
                    v[u] = ok ? p.in[(size_t)by * p.nx + gx] : 0.0f;
                }
                if (k < p.cols_l) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = r0 + wave + 4 * u;
                        if (r < p.rows_l) L[r * STRIDE + k] = v[u];
                    }
                }
            }
        }
    }
    __syncthreads();

    constexpr int S = ALONG_X ? 1 : STRIDE;
    float best[NA][OWN], centre[OWN];
    const float* Lw = ALONG_X ? L + lane * STRIDE + wave * OWN : L + wave * OWN * STRIDE + lane;
    const int self = -p.dj_min * STRIDE - p.di_min;
#pragma unroll
    for (int k = 0; k < OWN; ++k) {
        centre[k] = Lw[self + k * S] + p.height;
#pragma unroll
        for (int a = 0; a < NA; ++a) best[a][k] = -INFINITY;
    }
    for (int c = 0; c < p.n_cls; ++c) {
        float top[OWN];
#pragma unroll
        for (int k = 0; k < OWN; ++k) top[k] = -INFINITY;
        const int e8 = p.cls_first8[c + 1], e4 = p.cls_first4[c + 1], e2 = p.cls_first2[c + 1];
        for (int n = p.cls_first8[c]; n < e8; ++n) {
            const SxChain8 e = p.tab8[n];
            const float* q = Lw + e.off;
            float w[OWN + 7];
#pragma unroll
            for (int j = 0; j < OWN + 7; ++j) w[j] = q[j * S];
#pragma unroll
            for (int m = 0; m < 8; m += 2) {
#pragma unroll
                for (int k = 0; k < OWN; ++k) {
                    const float z0 = (w[m + k] - centre[k]) * e.inv[m];
                    const float z1 = (w[m + 1 + k] - centre[k]) * e.inv[m + 1];
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(top[k]) : "v"(z0), "v"(z1));
                }
            }
        }
        for (int n = p.cls_first4[c]; n < e4; ++n) {
            const SxChain4 e = p.tab4[n];
            const float* q = Lw + e.off;
            float w[OWN + 3];
#pragma unroll
            for (int j = 0; j < OWN + 3; ++j) w[j] = q[j * S];
#pragma unroll
            for (int m = 0; m < 4; m += 2) {
#pragma unroll
                for (int k = 0; k < OWN; ++k) {
                    const float z0 = (w[m + k] - centre[k]) * e.inv[m];
                    const float z1 = (w[m + 1 + k] - centre[k]) * e.inv[m + 1];
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(top[k]) : "v"(z0), "v"(z1));
                }
            }
        }
        for (int n = p.cls_first2[c]; n < e2; n += 4) {
            struct Quad {
                SxChain2 e[4];
            };
            const Quad g = *reinterpret_cast<const Quad*>(p.tab2 + n);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* q = Lw + g.e[u].off;
                float w[OWN + 1];
#pragma unroll
                for (int j = 0; j < OWN + 1; ++j) w[j] = q[j * S];
#pragma unroll
                for (int k = 0; k < OWN; ++k) {
                    const float z0 = (w[k] - centre[k]) * g.e[u].inv[0];
                    const float z1 = (w[1 + k] - centre[k]) * g.e[u].inv[1];
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(top[k]) : "v"(z0), "v"(z1));
                }
            }
        }
        const int mask = p.cls_mask[c];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            if (mask >> a & 1) {
#pragma unroll
                for (int k = 0; k < OWN; ++k) best[a][k] = fmaxf(best[a][k], top[k]);
            }
        }
    }
    const float rad2deg = 57.29577951308232f;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        if (a >= p.n_az) break;  // uniform
        const int w = p.window[a];
        if (ALONG_X) {
            // lanes run along y: each plane's tile goes through LDS so that the stores are row segments.  Lane l
            // then takes column l & 31 of rows 2 k + (l >> 5) of the wave's 16 rows.
            __syncthreads();
#pragma unroll
            for (int k = 0; k < OWN; ++k) L[lane * (TW + 1) + wave * OWN + k] = best[a][k];
            __syncthreads();
            const int col = lane & 31, half = lane >> 5;
#pragma unroll
            for (int k = 0; k < OWN; ++k) {
                const int row = wave * 16 + 2 * k + half;
                const float t = L[row * (TW + 1) + col];
                const int ox = ox0 + col, oy = oy0 + row;
                if (ox >= p.nx || oy >= p.out_row0 + p.out_rows) continue;
                const bool inside = oy >= w && oy < p.gny - w && ox >= w && ox < p.nx - w;
                float v = 0.0f;
                if (inside) v = t == -INFINITY ? NAN : atan_signed(t) * rad2deg;
                p.out[a][(size_t)(oy - p.out_row0) * p.nx + ox] = v;
            }
        } else {
            const int ox = ox0 + lane;
            if (ox < p.nx) {
#pragma unroll
                for (int k = 0; k < OWN; ++k) {
                    const int oy = oy0 + wave * OWN + k;
                    if (oy >= p.out_row0 + p.out_rows) continue;
                    const bool inside = oy >= w && oy < p.gny - w && ox >= w && ox < p.nx - w;
                    float v = 0.0f;
                    if (inside) v = best[a][k] == -INFINITY ? NAN : atan_signed(best[a][k]) * rad2deg;
                    p.out[a][(size_t)(oy - p.out_row0) * p.nx + ox] = v;
                }
            }
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
    p.out[o] = best == -INFINITY ? NAN : atan_signed(best) * 57.29577951308232f;
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
    if (c.ghost.armed && (pts.empty() || b.out_rows > kMaxLaunchRows)) return TOPO_AMD_EUNSUP;  // (not as one gated launch)
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
    a.rows_l = kSxTile + dj_max - dj_min;
    a.cols_l = kSxTile + di_max - di_min;
    const int stride = sx_stride_for(a.cols_l);
    // spare LDS behind the tile: a chain shorter than its table entry reads up to 7 samples past its run (7 rows
    // when the chains run along y)
    const size_t lds = (size_t)(a.rows_l + 8) * stride * sizeof(float);
    if (stride == 0 || lds > 160 * 1024) {
        if (c.ghost.armed) return TOPO_AMD_EUNSUP;  // the kernel without tiles knows no gate
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
    // chains along the direction that needs the fewest comparisons (padding included): down the columns, along the
    // rows, or along one of the two diagonals (whose tile is 15 columns wider: its own LDS stride)
    constexpr bool diag_on = true;
    constexpr int diag_min_saving = 5;
    // 32768^2, azimuth 45: radius 2000 m 428 -> 366 comparisons 28.4 -> 25.0 ms; 1000 m 146 -> 132: 8.45 -> 8.10;
    // 500 m 42 -> 36: 3.90 -> 3.99 (hence the floor below)
    constexpr int diag_min_cost = 100;
    const int cols_d = a.cols_l + kSxOwn - 1;
    const int stride_d = sx_stride_for(cols_d);
    const size_t lds_d = (size_t)(a.rows_l + 8) * stride_d * sizeof(float);
    const bool diag_fits = diag_on && stride_d != 0 && lds_d <= 160 * 1024;
    // pairs: any two chains of one length with the same weights, bit for bit
    constexpr bool pairs_on = true;
    auto pair_up = [&](const auto& all, auto& singles, auto& pairs, auto second) {
        std::vector<char> used(all.size(), 0);
        for (size_t i = 0; i < all.size(); ++i) {
            if (used[i]) continue;
            size_t partner = all.size();
            for (size_t k = i + 1; pairs_on && k < all.size(); ++k)
                if (!used[k] && std::memcmp(all[i].inv, all[k].inv, sizeof(all[i].inv)) == 0) {
                    partner = k;
                    break;
                }
            if (partner == all.size()) {
                singles.push_back(all[i]);
            } else {
                auto e = all[i];
                second(e) = all[partner].off;
                pairs.push_back(e);
                used[partner] = 1;
            }
        }
    };
    std::vector<SxChain8> t8, s8m[4], p8m[4];
    std::vector<SxChain4> t4, s4m[4], p4m[4];
    std::vector<SxChain2> t2, s2m[4], p2m[4];
    size_t cost[4];
    int mode = 0;
    for (int m = 0; m < 4; ++m) {
        if (m >= 2 && !diag_fits) break;
        sx_chains(pts, m, m >= 2 ? stride_d : stride, dj_min, di_min, &t8, &t4, &t2);
        pair_up(t8, s8m[m], p8m[m], [](SxChain8& e) -> int& { return e.pad[0]; });
        pair_up(t4, s4m[m], p4m[m], [](SxChain4& e) -> int& { return e.pad[0]; });
        pair_up(t2, s2m[m], p2m[m], [](SxChain2& e) -> int& { return e.pad; });
        // comparisons per pixel, padding included; a pair costs its 8 (4, 2) and a sample maximum per own pixel and chain
        cost[m] = 8 * s8m[m].size() + 4 * s4m[m].size() + 2 * s2m[m].size() + 10 * p8m[m].size() + 5 * p4m[m].size() + 3 * p2m[m].size();
        // a diagonal scan has to save comparisons to be worth its wider tile and its extra column of tiles
        // (TOPO_AMD_SX_DIAG_MIN_SAVING, percent), and the scan has to be long enough for that to show
        // (TOPO_AMD_SX_DIAG_MIN_COST comparisons)
        if (m > 0 && (m < 2 ? cost[m] < cost[mode]
                            : cost[mode] >= (size_t)diag_min_cost && 100 * cost[m] < (size_t)(100 - diag_min_saving) * cost[mode]))
            mode = m;
    }
    const bool along_x = mode == 1;
    const int diag = mode == 2 ? 1 : mode == 3 ? -1 : 0;
    const std::vector<SxChain8>&s8 = s8m[mode], &p8 = p8m[mode];
    const std::vector<SxChain4>&s4 = s4m[mode], &p4 = p4m[mode];
    std::vector<SxChain2>& s2 = s2m[mode];
    const std::vector<SxChain2>& p2 = p2m[mode];
    while (s2.size() % 4) {  // the kernel takes four at a time: entries whose products are all NaN
        SxChain2 e{};
        e.inv[0] = e.inv[1] = std::nanf("");
        s2.push_back(e);
    }
    const size_t scan_cost = cost[mode];
    const SxChain8 none8{};
    const SxChain4 none4{};
    const SxChain2 none2{};
    void *d_t8 = nullptr, *d_t4 = nullptr, *d_t2 = nullptr;  // (an empty table still uploads one entry; its count stays 0)
    void *d_p8 = nullptr, *d_p4 = nullptr, *d_p2 = nullptr;
    TOPO_TRY(upload_table(0, s8.empty() ? &none8 : s8.data(), std::max<size_t>(1, s8.size()) * sizeof(SxChain8), &d_t8));
    TOPO_TRY(upload_table(1, s2.empty() ? &none2 : s2.data(), std::max<size_t>(1, s2.size()) * sizeof(SxChain2), &d_t2));
    TOPO_TRY(upload_table(2, s4.empty() ? &none4 : s4.data(), std::max<size_t>(1, s4.size()) * sizeof(SxChain4), &d_t4));
    TOPO_TRY(upload_table(3, p8.empty() ? &none8 : p8.data(), std::max<size_t>(1, p8.size()) * sizeof(SxChain8), &d_p8));
    TOPO_TRY(upload_table(4, p2.empty() ? &none2 : p2.data(), std::max<size_t>(1, p2.size()) * sizeof(SxChain2), &d_p2));
    TOPO_TRY(upload_table(5, p4.empty() ? &none4 : p4.data(), std::max<size_t>(1, p4.size()) * sizeof(SxChain4), &d_p4));
    a.in = b.in;
    a.out = out;
    a.tab8 = (const SxChain8*)d_t8;
    a.tab4 = (const SxChain4*)d_t4;
    a.tab2 = (const SxChain2*)d_t2;
    a.n8 = (int)s8.size();
    a.n4 = (int)s4.size();
    a.n2 = (int)s2.size();
    a.tab8p = (const SxChain8*)d_p8;
    a.tab4p = (const SxChain4*)d_p4;
    a.tab2p = (const SxChain2*)d_p2;
    a.n8p = (int)p8.size();
    a.n4p = (int)p4.size();
    a.n2p = (int)p2.size();
    a.in_rows = b.in_rows;
    a.in_row0 = b.in_row0;
    a.gny = b.gny;
    a.nx = b.nx;
    a.out_row0 = b.out_row0;
    a.out_rows = b.out_rows;
    a.window = window;
    a.height = (float)height;
    // lanes along x and a long scan (radius 2000 m: 23.2 -> 22.0 ms; radius 500 m loses 6 %, 1000 m is even):
    // 128-row tiles with 8 waves while two such blocks share a CU
    int waves = 4;
    size_t lds_used = lds;
    a.tiles_y = a.n_top = a.n_bot = a.cleanup = 0;
    a.gate = Gate{nullptr, 0, nullptr, 0, nullptr, nullptr};
    // One launch, and for a row shard whose exchange is in flight (c.ghost.armed) its clean-up launch: the tile rows that
    // read ghost rows go last and wait at the gate (sx_kernel).  tile_h: rows per tile row; rows_l: rows a tile stages.
    auto go = [&](int use_stride, bool use_along_x, int use_waves, dim3 grid, size_t use_lds, int use_diag, int tile_h) -> int {
        a.tiles_y = (int)grid.y;
        if (!c.ghost.armed) return launch_sx_stride(use_stride, use_along_x, use_waves, grid, use_lds, c.compute, a, use_diag);
        const int T = (int)grid.y;
        int n_top = 0, n_bot = 0;
        for (int t = 0; t < T; ++t) {
            const int gy_first = b.out_row0 + t * tile_h + a.dj_min;
            const bool top = gy_first < c.ghost.ghost_lo, bot = gy_first + a.rows_l > c.ghost.ghost_hi;
            if (top && t != n_top) return TOPO_AMD_EUNSUP;  // (cannot happen: the tile rows go down the block)
            n_top += top ? 1 : 0;
            n_bot += bot ? 1 : 0;
        }
        if (n_top + n_bot > T || (size_t)(n_top + n_bot) * grid.x > c.ghost.slots) return TOPO_AMD_EUNSUP;
        // the blocks that wait at the gate hold their CU slots; RCCL's workgroups need room next to them, or nobody moves:
        // at most as many waiting blocks as leave one block slot per CU free (else: separate launches)
        int per_cu = 0;
        a.gate = c.ghost.gate;  // (from here on `a` names the gated instantiation of the kernel)
        TOPO_TRY(launch_sx_stride(use_stride, use_along_x, use_waves, grid, use_lds, c.compute, a, use_diag, &per_cu));
        if ((long)(n_top + n_bot) * grid.x > (long)(per_cu - 1) * c.num_cu) return TOPO_AMD_EUNSUP;
        a.n_top = n_top;
        a.n_bot = n_bot;
        c.ghost.armed = false;
        TOPO_TRY(launch_sx_stride(use_stride, use_along_x, use_waves, grid, use_lds, c.compute, a, use_diag));
        if (a.gate.errors == nullptr && n_top + n_bot > 0) {  // careful mode: the blocks that gave up at the gate
            TOPO_TRY(topo_amd_halo_wait());
            a.cleanup = 1;
            a.gate.word = c.ghost.gate.word;  // (non-null: the kernel's "this is a gated launch" switch)
            TOPO_TRY(launch_sx_stride(use_stride, use_along_x, use_waves, dim3(grid.x, (unsigned)(n_top + n_bot)), use_lds, c.compute, a,
                                      use_diag));
        }
        return TOPO_AMD_OK;
    };
    if (diag != 0) {
        a.cols_l = cols_d;
        dim3 dgrid((b.nx + kSxTile - 1) / kSxTile + 1, (b.out_rows + kSxTile - 1) / kSxTile);  // the slabs lean: one tile more
        return go(stride_d, false, 4, dgrid, lds_d, diag, kSxTile);
    }
    if (!along_x && b.out_rows >= 2 * kSxTile && scan_cost >= 256) {
        const size_t lds8 = (size_t)(a.rows_l + kSxTile + 8) * stride * sizeof(float);
        if (lds8 <= 80 * 1024) {
            waves = 8;
            a.rows_l += kSxTile;
            lds_used = lds8;
        }
    }
    const int span = kSxOwn * waves;
    dim3 grid((b.nx + kSxTile - 1) / kSxTile, (b.out_rows + span - 1) / span);
    return go(stride, along_x, waves, grid, lds_used, 0, along_x ? kSxTile : span);
}

namespace {

template <int I = 0>
int launch_sx_multi_stride(int stride, bool along_x, dim3 grid, size_t lds, hipStream_t stream, const SxMultiArgs& a) {
    if constexpr (I < kSxStrideCount) {
        if (stride == kSxStrides[I]) {
            auto go = [&](auto kernel) -> int {
                TOPO_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kernel, grid, dim3(kThreads), lds, stream, a);
                TOPO_HIP(hipGetLastError());
                return TOPO_AMD_OK;
            };
            constexpr int ST = kSxStrides[I];
            if (a.n_az <= 2) return along_x ? go(sx_multi_kernel<ST, true, 2>) : go(sx_multi_kernel<ST, false, 2>);
            if (a.n_az <= 4) return along_x ? go(sx_multi_kernel<ST, true, 4>) : go(sx_multi_kernel<ST, false, 4>);
            return along_x ? go(sx_multi_kernel<ST, true, kMaxAz>) : go(sx_multi_kernel<ST, false, kMaxAz>);
        }
        return launch_sx_multi_stride<I + 1>(stride, along_x, grid, lds, stream, a);
    } else {
        set_error("sx_multi: no kernel for LDS stride %d", stride);
        return TOPO_AMD_EUNSUP;
    }
}

}  // namespace

namespace {

struct SectorPoints {
    std::vector<std::pair<std::pair<int, int>, double>> pts;  // unique usable (dj, di), distance
    int dj_min = 0, dj_max = 0, di_min = 0, di_max = 0;       // box, the target pixel included
};

// LDS of the multi-azimuth tile for one orientation (0 when no compiled stride holds its columns)
size_t sx_multi_tile_bytes(int dj_min, int dj_max, int di_min, int di_max, bool along_x) {
    const int tw = along_x ? kSxMultiSpan : 64, th = along_x ? 64 : kSxMultiSpan;
    const int stride = sx_stride_for(tw + di_max - di_min);
    return stride ? (size_t)(th + dj_max - dj_min + 8) * stride * sizeof(float) : 0;
}
// what decides whether sectors share a launch: the larger of the two orientations (the launch picks one later)
size_t sx_tile_bytes(int dj_min, int dj_max, int di_min, int di_max) {
    const size_t v = sx_multi_tile_bytes(dj_min, dj_max, di_min, di_max, false);
    const size_t h = sx_multi_tile_bytes(dj_min, dj_max, di_min, di_max, true);
    return (v == 0 || h == 0) ? (size_t)1 << 30 : std::max(v, h);
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
    // classes of equal membership, each cut into chains along both axes; the axis with fewer comparisons wins
    std::vector<int> mask;
    std::vector<SxPoints> cls;
    for (size_t n = 0; n < all.size(); ++n) {
        if (n == 0 || all[n].second.first != all[n - 1].second.first) {
            mask.push_back(all[n].second.first);
            cls.emplace_back();
        }
        cls.back().push_back({all[n].first, all[n].second.second});
    }
    bool along_x = false;
    {
        size_t work[2] = {0, 0};
        std::vector<SxChain8> t8;
        std::vector<SxChain4> t4;
        std::vector<SxChain2> t2;
        for (int ax = 0; ax < 2; ++ax)
            for (auto& pts : cls) {
                sx_chains(pts, ax == 1, 1, a.dj_min, a.di_min, &t8, &t4, &t2);
                work[ax] += 8 * t8.size() + 4 * t4.size() + 2 * ((t2.size() + 3) / 4 * 4);
            }
        along_x = work[1] < work[0];
    }
    const int tw = along_x ? kSxMultiSpan : 64, th = along_x ? 64 : kSxMultiSpan;
    a.rows_l = th + dj_max - a.dj_min;
    a.cols_l = tw + di_max - a.di_min;
    const int stride = sx_stride_for(a.cols_l);
    TOPO_REQUIRE(stride != 0, "sx_multi: no kernel for a tile of %d columns", a.cols_l);
    const size_t lds = (size_t)(a.rows_l + 8) * stride * sizeof(float);
    std::vector<SxChain8> tab8;
    std::vector<SxChain4> tab4;
    std::vector<SxChain2> tab2;
    std::vector<int> first8, first4, first2;
    for (auto& pts : cls) {
        std::vector<SxChain8> t8;
        std::vector<SxChain4> t4;
        std::vector<SxChain2> t2;
        sx_chains(pts, along_x, stride, a.dj_min, a.di_min, &t8, &t4, &t2);
        while (t2.size() % 4) {
            SxChain2 e{};
            e.inv[0] = e.inv[1] = std::nanf("");
            t2.push_back(e);
        }
        first8.push_back((int)tab8.size());
        first4.push_back((int)tab4.size());
        first2.push_back((int)tab2.size());
        tab8.insert(tab8.end(), t8.begin(), t8.end());
        tab4.insert(tab4.end(), t4.begin(), t4.end());
        tab2.insert(tab2.end(), t2.begin(), t2.end());
    }
    first8.push_back((int)tab8.size());
    first4.push_back((int)tab4.size());
    first2.push_back((int)tab2.size());
    if (tab8.empty()) tab8.push_back(SxChain8{});
    if (tab4.empty()) tab4.push_back(SxChain4{});
    if (tab2.empty()) tab2.push_back(SxChain2{});
    // one table for the three lists of class boundaries (six table slots in all)
    const size_t nf = first8.size();
    std::vector<int> firsts(first8);
    firsts.insert(firsts.end(), first4.begin(), first4.end());
    firsts.insert(firsts.end(), first2.begin(), first2.end());
    void *d_t8 = nullptr, *d_t4 = nullptr, *d_t2 = nullptr, *d_f = nullptr, *d_mask = nullptr;
    TOPO_TRY(upload_table(0, tab8.data(), tab8.size() * sizeof(SxChain8), &d_t8));
    TOPO_TRY(upload_table(1, tab2.data(), tab2.size() * sizeof(SxChain2), &d_t2));
    TOPO_TRY(upload_table(2, firsts.data(), firsts.size() * sizeof(int), &d_f));
    TOPO_TRY(upload_table(3, tab4.data(), tab4.size() * sizeof(SxChain4), &d_t4));
    TOPO_TRY(upload_table(4, mask.data(), mask.size() * sizeof(int), &d_mask));
    a.in = b.in;
    for (int k = 0; k < kMaxAz; ++k) {
        a.out[k] = k < a1 - a0 ? outs[a0 + k] : nullptr;
        a.window[k] = k < a1 - a0 ? window[a0 + k] : 0;
    }
    a.tab8 = (const SxChain8*)d_t8;
    a.tab4 = (const SxChain4*)d_t4;
    a.tab2 = (const SxChain2*)d_t2;
    a.cls_first8 = (const int*)d_f;
    a.cls_first4 = (const int*)d_f + nf;
    a.cls_first2 = (const int*)d_f + 2 * nf;
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
    dim3 grid((b.nx + tw - 1) / tw, (b.out_rows + th - 1) / th);
    return launch_sx_multi_stride(stride, along_x, grid, lds, c.compute, a);
}

}  // namespace

int launch_sx_multi(const Block& b, int n_az, const int32_t* first, const int32_t* dj, const int32_t* di,
                    const double* dist, const int32_t* window, double height, float* const* outs) {
    // at most this much LDS per block, so that three blocks still share a CU
    constexpr size_t kGroupLds = 48 * 1024;
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
