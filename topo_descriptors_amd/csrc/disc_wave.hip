// K1/K2 fast path: disc sums by column runs + wavefront shift-accumulate, specialised per size.
//
// S(j, i) = sum over column offsets di of C(j, i + di),   C(j, c) = Q(j + hi(di) + 1, c) - Q(j + lo(di), c)
//
// where Q is the prefix sum down the columns of the staged tile and [lo(di), hi(di)] the
// vertical run of the disc at column offset di.  Lanes own NC = 4 adjacent columns, so one
// ds_read_b128 fetches a prefix row for all of them; the C values of the ~21 distinct runs of
// a 67-px disc sit in registers, and the sum over di is a chain of adds in which the partial
// sums hop one lane per step with a DPP wave shift (v_add_*_dpp wave_shl:1) - no LDS traffic
// and no shuffles for the 67 taps of the chain.  Per output pixel that is ~21 subtractions and
// ~67 additions instead of 134 LDS reads (row-prefix gather) or 3409 taps (direct).
//
// The kernel is instantiated per disc size (every run is a compile-time constant, which is
// what keeps the C values in statically indexed registers); sizes without an instantiation
// use the generic LDS kernel in disc.hip.
#include "common.hpp"
#include "disc_runs.hpp"

#include <cstdlib>

namespace topo {

namespace {

constexpr int NC = 4;          // columns per lane
constexpr int ROWW = 64 * NC;  // staged columns per tile row (one wave-row, 1 KiB)

constexpr int cdiv_floor(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
constexpr int cdiv_ceil(int a, int b) { return -cdiv_floor(-a, b); }

template <int SIZE>
struct Geo {
    static constexpr DiscTable<SIZE> T = make_disc_table<SIZE>();
    static constexpr int D_LO = cdiv_ceil(T.off_min - (NC - 1), NC);   // lane offsets spanned
    static constexpr int D_HI = cdiv_floor(T.off_max + (NC - 1), NC);
    static constexpr int NVL = 64 - (D_HI - D_LO);   // lanes that end up with a full sum
    static constexpr int TILE_W = NC * NVL;          // valid output columns per tile
    static constexpr int X0 = -D_LO * NC;            // staged column of the first valid output
    static_assert(NVL >= 8, "disc too wide for one wavefront of 4-column lanes");
};

struct WaveArgs {
    const float* in;
    float* tpi;
    float* sd;
    int in_rows, in_row0, gny, nx;
    int out_row0, out_rows;
    int debug;  // ablation switches, honoured only in -DTOPO_AMD_ABLATE profiling builds
    uint32_t* scratch;  // STD kernel: per-block planes for the per-row sums between its passes
};

#ifdef TOPO_AMD_ABLATE
#define ABLATE(p, bit) ((p).debug & (bit))
#else
#define ABLATE(p, bit) false
#endif

template <typename T>
struct alignas(16) Vec4 {
    T v[4];
};

#define DPP_WAVE_SHL1 0x130  // lane i takes lane i + 1; lane 63 takes 0 (bound_ctrl)

__device__ __forceinline__ float hop(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), DPP_WAVE_SHL1, 0xf, 0xf, true));
}
__device__ __forceinline__ int hop(int x) {
    return __builtin_amdgcn_update_dpp(0, x, DPP_WAVE_SHL1, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t hop(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, DPP_WAVE_SHL1, 0xf, 0xf, true);
}

// Disc sums of output row jj (tile-relative) for the NC output columns that end up in this
// lane: lane l receives the sums of staged columns NC * (l - D_LO) + t, valid for l < NVL.
// T is the type of the prefix sums (float, int32, uint32 with wrap-around).  HALF selects what
// of each uint32 column sum enters the chain: 0 all of it, 1 its low 16 bits, 2 its high 16 bits
// (two 16-bit chains give an exact 48-bit total where one uint32 chain could overflow).
template <int SIZE, typename T, int HALF = 0>
__device__ __forceinline__ void wave_disc_sum(const T* Q, int jj, int lane, T (&acc)[NC]) {
    using ACC = T;
    using G = Geo<SIZE>;
    constexpr int NR = G::T.num_runs;
    T cv[NR][NC];
    const T* col = Q + lane * NC;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const Vec4<T> top = *reinterpret_cast<const Vec4<T>*>(col + (jj + G::T.run_hi[r] - G::T.off_min + 1) * ROWW);
        const Vec4<T> bot = *reinterpret_cast<const Vec4<T>*>(col + (jj + G::T.run_lo[r] - G::T.off_min) * ROWW);
#pragma unroll
        for (int s = 0; s < NC; ++s) {
            T d = top.v[s] - bot.v[s];
            if (HALF == 1) d = (T)((uint32_t)d & 0xffffu);
            if (HALF == 2) d = (T)((uint32_t)d >> 16);
            cv[r][s] = d;
        }
    }
#pragma unroll
    for (int D = G::D_HI; D >= G::D_LO; --D) {
#pragma unroll
        for (int t = 0; t < NC; ++t) {
            // contributions of this lane's NC columns to output sub-column t: a short tree,
            // independent of the hop chain, so the chain itself is one dependent add per step
            ACC part = (ACC)0;
            bool any = false;
#pragma unroll
            for (int s = 0; s < NC; ++s) {
                const int di = NC * D + s - t;
                if (di >= G::T.off_min && di <= G::T.off_max) {
                    const ACC c = (ACC)cv[G::T.run_of[di - G::T.off_min]][s];
                    part = any ? part + c : c;
                    any = true;
                }
            }
            if (D == G::D_HI) {
                acc[t] = part;  // nothing to move before the first step
            } else if (any) {
                acc[t] = hop(acc[t]) + part;
            } else {
                acc[t] = hop(acc[t]);
            }
        }
    }
}

// One 16-byte row piece of the DEM, zero outside the global DEM (mode="same" padding).
// nx % 4 == 0 and gx % 4 == 0, so a float4 is entirely inside or entirely outside.  The load is
// unconditional (clamped address) so that all row loads of a tile can be in flight together.
__device__ __forceinline__ bool row4_inside(const WaveArgs& p, int gy, int gx) {
    const int by = gy - p.in_row0;
    return gy >= 0 && gy < p.gny && gx >= 0 && gx < p.nx && by >= 0 && by < p.in_rows;
}
__device__ __forceinline__ Vec4<float> load_row4(const WaveArgs& p, int gy, int gx) {
    const bool ok = row4_inside(p, gy, gx);
    const size_t idx = ok ? (size_t)(gy - p.in_row0) * p.nx + gx : 0;
    return *reinterpret_cast<const Vec4<float>*>(p.in + idx);
}

// ---- TPI alone: one float32 chain on a = x - c --------------------------------------------------
// c is integer-valued, so on integer-valued DEMs every partial sum is an integer below 2^24 and
// the float32 arithmetic is exact; on fractional DEMs it rounds at the 1e-5 m level after the
// division by n-1.
//
// Persistent blocks: each block walks over tiles (vertical neighbours first, XCD-contiguous so
// that the ghost rows two tiles share meet in one L2), and the DEM rows of the next tile are
// already in flight to registers while the current tile is being summed.
template <int SIZE, int TH, int NWAVES, bool PREFETCH>
__global__ __launch_bounds__(NWAVES * 64) void disc_wave_tpi_kernel(WaveArgs p, int tiles_x, int tiles_y) {
    using G = Geo<SIZE>;
    constexpr int NROWS = TH + SIZE - 1;              // staged DEM rows
    constexpr int SL = (NROWS + NWAVES - 1) / NWAVES;  // rows per wave in the staging phase
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Q = lds;                          // (NROWS + 1) x ROWW column prefix sums
    float* TOT = lds + (NROWS + 1) * ROWW;   // NWAVES x ROWW segment totals
    // border tiles only: per output row, prefix over the column offsets di of the number of
    // in-domain rows of the disc run at di (what the padded convolution really sums over)
    unsigned short* PL = reinterpret_cast<unsigned short*>(TOT + NWAVES * ROWW);  // TH x (SIZE + 1)

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ntiles = tiles_x * tiles_y;
    // blocks are dealt round-robin over the 8 XCDs: give each XCD a contiguous run of tiles
    const int nb = gridDim.x;
    const int per_xcd = nb >> 3;
    const int vb = (nb & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);

    const int out_lane = lane - G::D_LO;  // lane whose columns this lane's sums belong to
    // 1/(n-1): inf for size 1 -> non-finite results, like the reference's division by zero
    const double inv_nm1 = 1.0 / ((double)G::T.taps - 1.0);

    // row tiles are aligned to GLOBAL multiples of TH so that a pixel is computed by the same
    // instruction sequence whatever row block it belongs to
    const int ty0 = p.out_row0 / TH;
    auto tile_origin = [&](int tile, int& ox0, int& oy0) {
        ox0 = (tile / tiles_y) * G::TILE_W;
        oy0 = (ty0 + tile % tiles_y) * TH;
    };
    auto centre_index = [&](int ox0, int oy0) -> size_t {
        int cy = min(max(oy0 + TH / 2, 0), p.gny - 1);
        cy = min(max(cy, p.in_row0), p.in_row0 + p.in_rows - 1);
        const int cx = min(ox0 + G::TILE_W / 2, p.nx - 1);
        return (size_t)(cy - p.in_row0) * p.nx + cx;
    };

    Vec4<float> v[SL];
    float craw;
    int tile = vb;
    if (tile < ntiles) {
        int ox0, oy0;
        tile_origin(tile, ox0, oy0);
        craw = p.in[centre_index(ox0, oy0)];
#pragma unroll
        for (int k = 0; k < SL; ++k)
            v[k] = load_row4(p, oy0 + G::T.off_min + wave * SL + k, ox0 - G::X0 + lane * NC);
    }
    for (; tile < ntiles; tile += nb) {
        int ox0, oy0;
        tile_origin(tile, ox0, oy0);
        const int gx = ox0 - G::X0 + lane * NC;
        const int gy0 = oy0 + G::T.off_min;
        // integer offset near the local elevation
        float c = truncf(craw);
        if (!(fabsf(c) < 1.0e9f)) c = 0.0f;
        // tiles whose halo leaves the DEM: padded taps are staged as a = 0 and the number of
        // in-domain taps is counted per pixel instead (exact, no -c bulk in the sums)
        const bool border = !ABLATE(p, 4) && (gy0 < 0 || gy0 + NROWS > p.gny || ox0 - G::X0 < 0 || ox0 - G::X0 + ROWW > p.nx);

        // ---- phase 1: column prefix sums, each wave scans a segment of rows ------------------
        if (!ABLATE(p, 2)) {
            Vec4<float> run{{0.f, 0.f, 0.f, 0.f}};
            if (wave == 0) *reinterpret_cast<Vec4<float>*>(Q + lane * NC) = run;
#pragma unroll
            for (int k = 0; k < SL; ++k) {
                const int r = wave * SL + k;
                if (r < NROWS) {
                    const bool ok = row4_inside(p, gy0 + r, gx);
#pragma unroll
                    for (int s = 0; s < NC; ++s) run.v[s] += ok ? v[k].v[s] - c : (border ? 0.0f : -c);
                    *reinterpret_cast<Vec4<float>*>(Q + (r + 1) * ROWW + lane * NC) = run;
                }
            }
            *reinterpret_cast<Vec4<float>*>(TOT + wave * ROWW + lane * NC) = run;
        }
        // the rows of the next tile start their trip now and land during phase 2
        if (PREFETCH && tile + nb < ntiles && !ABLATE(p, 8)) {
            int nx0, ny0;
            tile_origin(tile + nb, nx0, ny0);
            craw = p.in[centre_index(nx0, ny0)];
#pragma unroll
            for (int k = 0; k < SL; ++k)
                v[k] = load_row4(p, ny0 + G::T.off_min + wave * SL + k, nx0 - G::X0 + lane * NC);
        }
        if (!ABLATE(p, 2)) {
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): LDS writes done, loads stay in flight
            __builtin_amdgcn_s_barrier();
            if (border && threadIdx.x < TH) {
                const int oy = oy0 + (int)threadIdx.x;
                unsigned short* row = PL + threadIdx.x * (SIZE + 1);
                int run = 0;
                row[0] = 0;
#pragma unroll 1
                for (int k = 0; k < SIZE; ++k) {
                    const int top = max(oy + G::T.lo[k], 0);
                    const int bot = min(oy + G::T.hi[k], p.gny - 1);
                    run += max(bot - top + 1, 0);
                    row[k + 1] = (unsigned short)run;
                }
            }
            if (wave > 0) {
                Vec4<float> off{{0.f, 0.f, 0.f, 0.f}};
                for (int w = 0; w < wave; ++w) {
                    const Vec4<float> t = *reinterpret_cast<const Vec4<float>*>(TOT + w * ROWW + lane * NC);
#pragma unroll
                    for (int s = 0; s < NC; ++s) off.v[s] += t.v[s];
                }
#pragma unroll
                for (int k = 0; k < SL; ++k) {
                    const int r = wave * SL + k;
                    if (r < NROWS) {
                        Vec4<float>* q = reinterpret_cast<Vec4<float>*>(Q + (r + 1) * ROWW + lane * NC);
                        Vec4<float> x = *q;
#pragma unroll
                        for (int s = 0; s < NC; ++s) x.v[s] += off.v[s];
                        *q = x;
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
        }

        // ---- phase 2: each wave takes every NWAVES-th output row --------------------------------
        const int ocol = ox0 + lane * NC;  // global column of acc[0] (lane < NVL)
        const bool lane_ok = lane < G::NVL && ocol < p.nx;
#pragma unroll 1
        for (int jj = wave; jj < TH; jj += NWAVES) {
            float acc[NC];
            if (ABLATE(p, 1)) {
                acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
            } else {
                wave_disc_sum<SIZE, float>(Q, jj, lane, acc);
            }
            const int oy = oy0 + jj;
            if (lane_ok && oy >= p.out_row0 && oy < p.out_row0 + p.out_rows) {
                // the pixel itself and the zeroed tap, recovered from the prefix rows
                const float* selfp = Q + (jj - G::T.off_min) * ROWW + out_lane * NC;
                const Vec4<float> s1 = *reinterpret_cast<const Vec4<float>*>(selfp + ROWW);
                const Vec4<float> s0 = *reinterpret_cast<const Vec4<float>*>(selfp);
                Vec4<float> res;
#pragma unroll
                for (int t = 0; t < NC; ++t) {
                    const float self = s1.v[t] - s0.v[t];
                    float ctr = self;
                    if (G::T.centre != 0) {
                        const float* cp = Q + (jj + G::T.centre - G::T.off_min) * ROWW + out_lane * NC + t + G::T.centre;
                        ctr = cp[ROWW] - cp[0];
                    }
                    if (!border) {
                        res.v[t] = (float)((double)self - (double)(acc[t] - ctr) * inv_nm1);
                    } else {
                        // sum over in-domain taps of x = acc + c m; the zeroed tap contributes
                        // x_ctr only when it lies inside the DEM
                        // in-domain taps: column offsets [d_lo, d_hi] keep ox + di inside the DEM
                        const int d_lo = max(G::T.off_min, -(ocol + t));
                        const int d_hi = min(G::T.off_max, p.nx - 1 - (ocol + t));
                        const unsigned short* pl = PL + jj * (SIZE + 1) - G::T.off_min;
                        const int m = d_hi >= d_lo ? (int)pl[d_hi + 1] - (int)pl[d_lo] : 0;
                        const int cy2 = oy + G::T.centre, cx2 = ocol + t + G::T.centre;
                        const bool ctr_in = cy2 >= 0 && cy2 < p.gny && cx2 >= 0 && cx2 < p.nx;
                        const double x_ctr = ctr_in ? (double)ctr + (double)c : 0.0;
                        const double total = (double)acc[t] + (double)c * (double)m;
                        res.v[t] = (float)((double)self + (double)c - (total - x_ctr) * inv_nm1);
                    }
                }
                *reinterpret_cast<Vec4<float>*>(p.tpi + (size_t)(oy - p.out_row0) * p.nx + ocol) = res;
            }
        }
        if (!PREFETCH && tile + nb < ntiles) {
            int nx0, ny0;
            tile_origin(tile + nb, nx0, ny0);
            craw = p.in[centre_index(nx0, ny0)];
#pragma unroll
            for (int k = 0; k < SL; ++k)
                v[k] = load_row4(p, ny0 + G::T.off_min + wave * SL + k, nx0 - G::X0 + lane * NC);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();  // every wave is done reading Q before it is overwritten
    }
}

template <int SIZE, int TH, int NWAVES, bool PREFETCH>
int launch_wave_tpi(const Block& b, float* tpi_out) {
    using G = Geo<SIZE>;
    Context& c = ctx();
    static const int debug = getenv("TOPO_AMD_DEBUG") ? atoi(getenv("TOPO_AMD_DEBUG")) : 0;
    WaveArgs a{b.in, tpi_out, nullptr, b.in_rows, b.in_row0, b.gny, b.nx, b.out_row0, b.out_rows, debug, nullptr};
    constexpr size_t lds = (size_t)((TH + SIZE) + NWAVES) * ROWW * sizeof(float) + (size_t)TH * (SIZE + 1) * sizeof(unsigned short);
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    static_assert(SIZE * SIZE < 65536, "tap counts must fit the 16-bit border table");
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        TOPO_HIP(hipFuncSetAttribute((const void*)disc_wave_tpi_kernel<SIZE, TH, NWAVES, PREFETCH>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int n = 0;
        TOPO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &n, (const void*)disc_wave_tpi_kernel<SIZE, TH, NWAVES, PREFETCH>, NWAVES * 64, lds));
        blocks_per_cu = n < 1 ? 1 : n;
    }
    const int tiles_x = (b.nx + G::TILE_W - 1) / G::TILE_W;
    const int tiles_y = (b.out_row0 + b.out_rows - 1) / TH - b.out_row0 / TH + 1;
    const long ntiles = (long)tiles_x * tiles_y;
    long grid = (long)c.num_cu * blocks_per_cu;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((disc_wave_tpi_kernel<SIZE, TH, NWAVES, PREFETCH>), dim3((unsigned)grid), dim3(NWAVES * 64),
                       lds, c.compute, a, tiles_x, tiles_y);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

// ---- STD, and TPI + STD fused: exact integer chains ---------------------------------------------
// With u = trunc(x) - c, f = x - trunc(x) and the sums taken over the in-domain taps (m of them):
//   s1 = sum x          = Su + Sf + c m
//   s2 = sum trunc(x)^2 = Su2 + 2 c Su + c^2 m            (the int32 quirk of topo.py:300)
//   STD = sqrt(max(0, (s2 - s1^2/n) / (n-1))),   TPI = x - (s1 - x_ctr) / (n-1)
// Su and Su2 are integer sums (int32 / uint32 prefix sums cannot round; the uint32 chain switches
// to two 16-bit half chains when a tile's |u| would let 3409 u^2 pass 2^32), Sf is a small float
// sum that exists only on fractional DEMs.  Tiles holding non-finite or absurd samples (|u| so
// large that one 67-row column sum of u^2 passes 2^32, e.g. -9999 nodata next to real terrain)
// run float chains on a = x - c and (trunc(x) - c)^2 instead, so NaN propagates and nothing wraps.
enum Stage { kStU = 0, kStU2 = 1, kStF = 2, kStA = 3, kStT2 = 4 };
enum TileFlags { kTileFrac = 1, kTileWide = 2, kTileFloat = 4 };

template <int WHAT>
__device__ __forceinline__ uint32_t stage_value(float x, float c, int ci) {
    if (WHAT == kStA) return __float_as_uint(x - c);
    const float t = truncf(x);
    if (WHAT == kStF) return __float_as_uint(x - t);
    if (WHAT == kStT2) {
        const float u = t - c;
        return __float_as_uint(u * u);
    }
    const int u = (int)t - ci;
    if (WHAT == kStU) return (uint32_t)u;
    return (uint32_t)u * (uint32_t)u;
}

// All waves: load the tile's rows, transform, and leave the column prefix sums of it in Q.
// Padded (out-of-domain) samples are staged as 0; the caller accounts for them through m.
template <int SIZE, int TH, int NWAVES, int WHAT, typename T>
__device__ __forceinline__ int stage_prefix(const WaveArgs& p, uint32_t* lds, int gy0, int gx, float c,
                                            int ci, float lim32, float limcv) {
    constexpr int NROWS = TH + SIZE - 1;
    constexpr int SL = (NROWS + NWAVES - 1) / NWAVES;
    T* Q = reinterpret_cast<T*>(lds);
    T* TOT = Q + (NROWS + 1) * ROWW;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    Vec4<float> v[SL];
#pragma unroll
    for (int k = 0; k < SL; ++k) v[k] = load_row4(p, gy0 + wave * SL + k, gx);
    int flags = 0;
    Vec4<T> run{{(T)0, (T)0, (T)0, (T)0}};
    if (wave == 0) *reinterpret_cast<Vec4<T>*>(Q + lane * NC) = run;
#pragma unroll
    for (int k = 0; k < SL; ++k) {
        const int r = wave * SL + k;
        if (r < NROWS) {
            const bool ok = row4_inside(p, gy0 + r, gx);
#pragma unroll
            for (int s = 0; s < NC; ++s) {
                const float x = v[k].v[s];
                if (WHAT == kStU && ok) {
                    const float t = truncf(x);
                    const float au = fabsf(t - c);
                    if (x != t) flags |= kTileFrac;
                    if (!(au <= lim32)) flags |= kTileWide;
                    if (!(au <= limcv)) flags |= kTileFloat;  // also NaN / inf
                }
                const uint32_t bits = ok ? stage_value<WHAT>(x, c, ci) : 0u;
                T val;
                __builtin_memcpy(&val, &bits, sizeof(T));
                run.v[s] += val;
            }
            *reinterpret_cast<Vec4<T>*>(Q + (r + 1) * ROWW + lane * NC) = run;
        }
    }
    *reinterpret_cast<Vec4<T>*>(TOT + wave * ROWW + lane * NC) = run;
    const int all = __syncthreads_or(flags);
    if (wave > 0) {
        Vec4<T> off{{(T)0, (T)0, (T)0, (T)0}};
        for (int w = 0; w < wave; ++w) {
            const Vec4<T> t = *reinterpret_cast<const Vec4<T>*>(TOT + w * ROWW + lane * NC);
#pragma unroll
            for (int s = 0; s < NC; ++s) off.v[s] += t.v[s];
        }
#pragma unroll
        for (int k = 0; k < SL; ++k) {
            const int r = wave * SL + k;
            if (r < NROWS) {
                Vec4<T>* q = reinterpret_cast<Vec4<T>*>(Q + (r + 1) * ROWW + lane * NC);
                Vec4<T> x = *q;
#pragma unroll
                for (int s = 0; s < NC; ++s) x.v[s] += off.v[s];
                *q = x;
            }
        }
    }
    __syncthreads();
    return all;
}

template <int SIZE, int TH, int NWAVES, bool WANT_TPI, bool WANT_STD>
__global__ __launch_bounds__(NWAVES * 64) void disc_wave_std_kernel(WaveArgs p, int tiles_x, int tiles_y) {
    using G = Geo<SIZE>;
    constexpr int NROWS = TH + SIZE - 1;
    constexpr int RW = TH / NWAVES;  // output rows per wave
    static_assert(TH % NWAVES == 0, "rows must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    unsigned short* PL = reinterpret_cast<unsigned short*>(lds_u + (NROWS + 1 + NWAVES) * ROWW);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ntiles = tiles_x * tiles_y;
    const int nb = gridDim.x;
    const int per_xcd = nb >> 3;
    const int vb = (nb & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const double n = (double)G::T.taps;
    const double inv_nm1 = 1.0 / (n - 1.0);
    const float lim32 = floorf(sqrtf(4294967295.0f / (float)G::T.taps));
    const float limcv = fminf(46000.0f, floorf(sqrtf(4294967295.0f / (float)SIZE)));

    for (int tile = vb; tile < ntiles; tile += nb) {
        const int ox0 = (tile / tiles_y) * G::TILE_W;
        const int oy0 = (p.out_row0 / TH + tile % tiles_y) * TH;  // global multiples of TH
        const int gx = ox0 - G::X0 + lane * NC;
        const int gy0 = oy0 + G::T.off_min;
        int cy = min(max(oy0 + TH / 2, 0), p.gny - 1);
        cy = min(max(cy, p.in_row0), p.in_row0 + p.in_rows - 1);
        const int cx = min(ox0 + G::TILE_W / 2, p.nx - 1);
        float c = truncf(p.in[(size_t)(cy - p.in_row0) * p.nx + cx]);
        if (!(fabsf(c) < 1.0e9f)) c = 0.0f;
        const int ci = (int)c;
        const bool border = gy0 < 0 || gy0 + NROWS > p.gny || ox0 - G::X0 < 0 || ox0 - G::X0 + ROWW > p.nx;

        // per-row sums live in this block's scratch planes between the passes (registers cannot
        // hold RW x NC x 4 values next to the chain): plane 0 Su (int32) or Sa (float bits),
        // 1/2 Su2 low/high word or St2 (float bits in 1), 3 Sf.  Written and read by the same lane.
        uint32_t* plane = p.scratch + (size_t)blockIdx.x * (4 * TH * ROWW) + lane * NC;
        auto put = [&](int which, int jj, const uint32_t (&val)[NC]) {
            Vec4<uint32_t> x{{val[0], val[1], val[2], val[3]}};
            *reinterpret_cast<Vec4<uint32_t>*>(plane + (which * TH + jj) * ROWW) = x;
        };
        auto get = [&](int which, int jj) {
            return *reinterpret_cast<const Vec4<uint32_t>*>(plane + (which * TH + jj) * ROWW);
        };

        const int flags = stage_prefix<SIZE, TH, NWAVES, kStU, int>(p, lds_u, gy0, gx, c, ci, lim32, limcv);
        const bool use_float = (flags & kTileFloat) != 0;
        const bool wide = (flags & kTileWide) != 0;
        const bool frac = (flags & kTileFrac) != 0;
        if (border && threadIdx.x < TH) {
            // in-domain rows of each column run, prefix over the column offsets (see TPI kernel)
            const int oy = oy0 + (int)threadIdx.x;
            unsigned short* row = PL + threadIdx.x * (SIZE + 1);
            int run = 0;
            row[0] = 0;
#pragma unroll 1
            for (int k = 0; k < SIZE; ++k) {
                const int top = max(oy + G::T.lo[k], 0);
                const int bot = min(oy + G::T.hi[k], p.gny - 1);
                run += max(bot - top + 1, 0);
                row[k + 1] = (unsigned short)run;
            }
        }
        if (!use_float) {
#pragma unroll 1
            for (int k = 0; k < RW; ++k) {
                int acc[NC];
                wave_disc_sum<SIZE, int>(reinterpret_cast<const int*>(lds_u), wave + k * NWAVES, lane, acc);
                const uint32_t bits[NC] = {(uint32_t)acc[0], (uint32_t)acc[1], (uint32_t)acc[2], (uint32_t)acc[3]};
                put(0, wave + k * NWAVES, bits);
            }
            if (WANT_STD) {
            __syncthreads();
            stage_prefix<SIZE, TH, NWAVES, kStU2, uint32_t>(p, lds_u, gy0, gx, c, ci, lim32, limcv);
            if (!wide) {
#pragma unroll 1
                for (int k = 0; k < RW; ++k) {
                    uint32_t acc[NC];
                    wave_disc_sum<SIZE, uint32_t>(lds_u, wave + k * NWAVES, lane, acc);
                    put(1, wave + k * NWAVES, acc);
                }
            } else {
#pragma unroll 1
                for (int k = 0; k < RW; ++k) {
                    uint32_t lo[NC], hi[NC];  // sums of the low / high 16 bits of the column sums
                    wave_disc_sum<SIZE, uint32_t, 1>(lds_u, wave + k * NWAVES, lane, lo);
                    wave_disc_sum<SIZE, uint32_t, 2>(lds_u, wave + k * NWAVES, lane, hi);
                    put(1, wave + k * NWAVES, lo);
                    put(2, wave + k * NWAVES, hi);
                }
            }
            }
        } else {
            __syncthreads();
            stage_prefix<SIZE, TH, NWAVES, kStA, float>(p, lds_u, gy0, gx, c, ci, lim32, limcv);
#pragma unroll 1
            for (int k = 0; k < RW; ++k) {
                float acc[NC];
                wave_disc_sum<SIZE, float>(reinterpret_cast<const float*>(lds_u), wave + k * NWAVES, lane, acc);
                const uint32_t bits[NC] = {__float_as_uint(acc[0]), __float_as_uint(acc[1]),
                                           __float_as_uint(acc[2]), __float_as_uint(acc[3])};
                put(0, wave + k * NWAVES, bits);
            }
            if (WANT_STD) {
            __syncthreads();
            stage_prefix<SIZE, TH, NWAVES, kStT2, float>(p, lds_u, gy0, gx, c, ci, lim32, limcv);
#pragma unroll 1
            for (int k = 0; k < RW; ++k) {
                float acc[NC];
                wave_disc_sum<SIZE, float>(reinterpret_cast<const float*>(lds_u), wave + k * NWAVES, lane, acc);
                const uint32_t bits[NC] = {__float_as_uint(acc[0]), __float_as_uint(acc[1]),
                                           __float_as_uint(acc[2]), __float_as_uint(acc[3])};
                put(1, wave + k * NWAVES, bits);
            }
            }
        }
        if (frac) {
            __syncthreads();
            stage_prefix<SIZE, TH, NWAVES, kStF, float>(p, lds_u, gy0, gx, c, ci, lim32, limcv);
#pragma unroll 1
            for (int k = 0; k < RW; ++k) {
                float acc[NC];
                wave_disc_sum<SIZE, float>(reinterpret_cast<const float*>(lds_u), wave + k * NWAVES, lane, acc);
                const uint32_t bits[NC] = {__float_as_uint(acc[0]), __float_as_uint(acc[1]),
                                           __float_as_uint(acc[2]), __float_as_uint(acc[3])};
                put(3, wave + k * NWAVES, bits);
            }
        }

        // ---- finalise in float64 -----------------------------------------------------------------
        const int ocol = ox0 + lane * NC;
        const bool lane_ok = lane < G::NVL && ocol < p.nx;
#pragma unroll 1
        for (int k = 0; k < RW; ++k) {
            const int jj = wave + k * NWAVES;
            const int oy = oy0 + jj;
            if (!lane_ok || oy < p.out_row0 || oy >= p.out_row0 + p.out_rows) continue;
            Vec4<float> xs{{0.f, 0.f, 0.f, 0.f}};
            if (WANT_TPI) xs = *reinterpret_cast<const Vec4<float>*>(p.in + (size_t)(oy - p.in_row0) * p.nx + ocol);
            const Vec4<uint32_t> q0 = get(0, jj);
            Vec4<uint32_t> q1{{0u, 0u, 0u, 0u}};
            if (WANT_STD) q1 = get(1, jj);
            Vec4<uint32_t> q2{{0u, 0u, 0u, 0u}}, q3{{0u, 0u, 0u, 0u}};
            if (WANT_STD && wide && !use_float) q2 = get(2, jj);
            if (frac) q3 = get(3, jj);
            Vec4<float> out_t, out_s;
#pragma unroll
            for (int t = 0; t < NC; ++t) {
                double m = n;
                if (border) {
                    const int d_lo = max(G::T.off_min, -(ocol + t));
                    const int d_hi = min(G::T.off_max, p.nx - 1 - (ocol + t));
                    const unsigned short* pl = PL + jj * (SIZE + 1) - G::T.off_min;
                    m = d_hi >= d_lo ? (double)((int)pl[d_hi + 1] - (int)pl[d_lo]) : 0.0;
                }
                const double sf = (double)__uint_as_float(q3.v[t]);
                double su, su2;  // sums of u and u^2 over the in-domain taps
                if (!use_float) {
                    su = (double)(int)q0.v[t];
                    su2 = (double)q2.v[t] * 65536.0 + (double)q1.v[t];  // q2 = 0 unless the tile is wide
                } else {
                    su = (double)__uint_as_float(q0.v[t]) - sf;  // Sa = Su + Sf
                    su2 = (double)__uint_as_float(q1.v[t]);
                }
                const double cd = (double)c;
                // su + c m is an exact integer (= sum of trunc(x)), so the result does not depend on
                // which c the tile happened to use
                const double s1 = (su + cd * m) + sf;
                if (WANT_STD) {
                    const double s2 = su2 + 2.0 * cd * su + cd * cd * m;
                    double var = (s2 - s1 * s1 / n) * inv_nm1;
                    if (var < 0.0) var = 0.0;  // keeps NaN, like np.clip
                    out_s.v[t] = (float)sqrt(var);
                }
                if (WANT_TPI) {
                    const int cy2 = oy + G::T.centre, cx2 = ocol + t + G::T.centre;
                    double x_ctr = (double)xs.v[t];
                    if (G::T.centre != 0) {
                        const bool in = cy2 >= 0 && cy2 < p.gny && cx2 >= 0 && cx2 < p.nx;
                        x_ctr = in ? (double)p.in[(size_t)(cy2 - p.in_row0) * p.nx + cx2] : 0.0;
                    }
                    out_t.v[t] = (float)((double)xs.v[t] - (s1 - x_ctr) * inv_nm1);
                }
            }
            const size_t o = (size_t)(oy - p.out_row0) * p.nx + ocol;
            if (WANT_STD) *reinterpret_cast<Vec4<float>*>(p.sd + o) = out_s;
            if (WANT_TPI) *reinterpret_cast<Vec4<float>*>(p.tpi + o) = out_t;
        }
        __syncthreads();  // Q and PL are rewritten by the next tile
    }
}

template <int SIZE, int TH, int NWAVES, bool WANT_TPI, bool WANT_STD>
int launch_wave_std(const Block& b, float* tpi_out, float* std_out) {
    using G = Geo<SIZE>;
    Context& c = ctx();
    WaveArgs a{b.in, tpi_out, std_out, b.in_rows, b.in_row0, b.gny, b.nx, b.out_row0, b.out_rows, 0, nullptr};
    constexpr size_t lds = (size_t)((TH + SIZE) + NWAVES) * ROWW * sizeof(float) +
                           (size_t)TH * (SIZE + 1) * sizeof(unsigned short);
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    static_assert(SIZE * SIZE < 65536, "tap counts must fit the 16-bit border table");
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        TOPO_HIP(hipFuncSetAttribute((const void*)disc_wave_std_kernel<SIZE, TH, NWAVES, WANT_TPI, WANT_STD>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int nblk = 0;
        TOPO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &nblk, (const void*)disc_wave_std_kernel<SIZE, TH, NWAVES, WANT_TPI, WANT_STD>, NWAVES * 64, lds));
        blocks_per_cu = nblk < 1 ? 1 : nblk;
    }
    const int tiles_x = (b.nx + G::TILE_W - 1) / G::TILE_W;
    const int tiles_y = (b.out_row0 + b.out_rows - 1) / TH - b.out_row0 / TH + 1;
    const long ntiles = (long)tiles_x * tiles_y;
    long grid = (long)c.num_cu * blocks_per_cu;
    if (grid > ntiles) grid = ntiles;
    void* scratch = nullptr;
    TOPO_TRY(workspace(2, (size_t)grid * 4 * TH * ROWW * sizeof(uint32_t), &scratch));
    a.scratch = (uint32_t*)scratch;
    hipLaunchKernelGGL((disc_wave_std_kernel<SIZE, TH, NWAVES, WANT_TPI, WANT_STD>), dim3((unsigned)grid),
                       dim3(NWAVES * 64), lds, c.compute, a, tiles_x, tiles_y);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

template <int SIZE>
int launch_wave_std_any(const Block& b, float* tpi_out, float* std_out) {
    if (tpi_out && std_out) return launch_wave_std<SIZE, 64, 8, true, true>(b, tpi_out, std_out);
    if (std_out) return launch_wave_std<SIZE, 64, 8, false, true>(b, tpi_out, std_out);
    return launch_wave_std<SIZE, 64, 8, true, false>(b, tpi_out, std_out);
}

}  // namespace

// Returns TOPO_AMD_EUNSUP when no specialisation covers the request (caller falls back to the
// generic kernel): needs nx % 4 == 0 for the 16-byte row accesses.
int launch_tpi_wave(const Block& b, int size, float* tpi_out) {
    // 12 waves x 5 rows per tile measured best on MI355X (8 x 8: +9 %, 16 x 4 without the
    // register prefetch: +70 %)
    constexpr int WAVES = 12;
    if (b.nx % 4 != 0 || (reinterpret_cast<uintptr_t>(b.in) & 15) || (reinterpret_cast<uintptr_t>(tpi_out) & 15))
        return TOPO_AMD_EUNSUP;
    switch (size) {
        case 67: return launch_wave_tpi<67, 60, WAVES, true>(b, tpi_out);
        case 65: return launch_wave_tpi<65, 60, WAVES, true>(b, tpi_out);
        case 17: return launch_wave_tpi<17, 60, WAVES, true>(b, tpi_out);
        case 7: return launch_wave_tpi<7, 60, WAVES, true>(b, tpi_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

// STD alone (tpi_out == NULL) or TPI + STD fused.
int launch_std_wave(const Block& b, int size, float* tpi_out, float* std_out) {
    if (b.nx % 4 != 0 || (reinterpret_cast<uintptr_t>(b.in) & 15) || (reinterpret_cast<uintptr_t>(std_out) & 15) ||
        (reinterpret_cast<uintptr_t>(tpi_out) & 15))
        return TOPO_AMD_EUNSUP;
    switch (size) {
        case 67: return launch_wave_std_any<67>(b, tpi_out, std_out);
        case 65: return launch_wave_std_any<65>(b, tpi_out, std_out);
        case 17: return launch_wave_std_any<17>(b, tpi_out, std_out);
        case 7: return launch_wave_std_any<7>(b, tpi_out, std_out);
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
