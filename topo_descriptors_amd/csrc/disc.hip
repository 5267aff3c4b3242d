// K1/K2: disc sliding-window sums -> TPI and STD.
//
// Replaces scipy.signal.convolve(dem, circular_kernel(size), "same") and the arithmetic
// around it in the reference (topo.py:168-181 for TPI, topo.py:295-307 for STD).
//
// Generic kernel (any radius whose tile fits LDS): a tile of the DEM plus its halo is staged
// in LDS, offset by an integer constant c close to the local elevation, turned into
// per-row prefix sums in place, and every output pixel then gathers one prefix difference
// per disc row (2 LDS reads per row of the disc instead of one per tap).
//
//   TPI  = a(j,i) - (sum_disc a - a(centre tap)) / (n-1)            a = x - c
//   STD  = sqrt(max(0, (sum u^2 - (sum a)^2/n - 2 c sum f) / (n-1)))  u = trunc(x) - c, f = x - trunc(x)
//
// which is the reference's (s2 - s1^2/n)/(n-1) with s1 = sum x, s2 = sum trunc(x)^2 (the int32
// truncation quirk of topo.py:300) rewritten around c so that float32 sums keep their digits.
// Out-of-domain taps read x = 0 (zero padding of mode="same"), n is always the full tap count.
#include "common.hpp"

#include <cstdlib>

namespace topo {

namespace {

constexpr int kThreads = 256;
constexpr int kTileW = 128;  // output columns per block; lanes run along columns (coalesced)

struct DiscArgs {
    const float* in;
    float* tpi;
    float* sd;
    const int* runs;  // per disc row: lo | (hi << 16), both biased by -di_min (>= 0)
    int in_rows, in_row0, gny, nx;
    int out_row0, out_rows;
    int tile_h;
    int n_disc_rows;   // dj_max - dj_min + 1
    int dj_min, di_min;
    int halo_cols;     // di_max - di_min
    int centre_dj, centre_di;
    int taps;
};

enum Pass { kPassA = 0, kPassU = 1, kPassU2 = 2, kPassF = 3, kPassT2 = 4 };
enum Flags { kFlagFrac = 1, kFlagBad = 2 };

__device__ __forceinline__ float load_padded(const DiscArgs& p, int gy, int gx) {
    // zero padding outside the global DEM; rows outside the block only feed unused outputs
    const int by = gy - p.in_row0;
    if (gy < 0 || gy >= p.gny || gx < 0 || gx >= p.nx || by < 0 || by >= p.in_rows) return 0.0f;
    return p.in[(size_t)by * p.nx + gx];
}

// What one staging pass writes to LDS for a DEM sample v (c is an integer-valued offset):
//   A   float  v - c                      U   int32  trunc(v) - c
//   U2  uint32 (trunc(v) - c)^2           F   float  v - trunc(v)
//   T2  float  (trunc(v) - c)^2           (float fallback of U2)
template <int PASS>
__device__ __forceinline__ uint32_t transform(float v, float c, int ci) {
    if (PASS == kPassA) return __float_as_uint(v - c);
    const float t = truncf(v);
    if (PASS == kPassF) return __float_as_uint(v - t);
    if (PASS == kPassT2) {
        const float u = t - c;
        return __float_as_uint(u * u);
    }
    const int u = (int)t - ci;
    if (PASS == kPassU) return (uint32_t)u;
    return (uint32_t)u * (uint32_t)u;
}

// Stage one transformed tile and turn every LDS row into a prefix sum in place:
// L[r][0] = 0, L[r][k] = sum of the first k staged values of row r (T = float or integer).
// Returns block-wide flags: fractional elevations present / samples the exact integer
// pipeline cannot take (non-finite, or |trunc(v) - c| > ulim).
template <int PASS, typename T>
__device__ int stage_and_scan(const DiscArgs& p, uint32_t* L, int stride, int rows_l, int cols_v,
                              int gy0, int gx0, float c, int ci, float ulim) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int flags = 0;
    for (int r = wave; r < rows_l; r += kThreads / 64) {
        const int gy = gy0 + r;
        uint32_t* row = L + r * stride + 1;
        for (int k = lane; k < cols_v; k += 64) {
            const float v = load_padded(p, gy, gx0 + k);
            if (PASS == kPassU) {
                const float t = truncf(v);
                if (v != t) flags |= kFlagFrac;
                if (!(fabsf(t - c) <= ulim)) flags |= kFlagBad;  // also catches NaN / inf
            }
            row[k] = transform<PASS>(v, c, ci);
        }
    }
    int all = 0;
    if (PASS == kPassU) {
        // __syncthreads_or is a LOGICAL or: reduce every flag bit on its own
        all |= __syncthreads_or(flags & kFlagFrac) ? kFlagFrac : 0;
        all |= __syncthreads_or(flags & kFlagBad) ? kFlagBad : 0;
    } else {
        __syncthreads();
    }
    for (int r = threadIdx.x; r < rows_l; r += kThreads) {
        T* row = reinterpret_cast<T*>(L + r * stride);
        T run = (T)0;
        row[0] = (T)0;
        int k = 1;
        for (; k + 3 <= cols_v; k += 4) {
            const T v0 = row[k], v1 = row[k + 1], v2 = row[k + 2], v3 = row[k + 3];
            const T s0 = run + v0;
            const T s1 = s0 + v1;
            const T s2 = s1 + v2;
            run = s2 + v3;
            row[k] = s0;
            row[k + 1] = s1;
            row[k + 2] = s2;
            row[k + 3] = run;
        }
        for (; k <= cols_v; ++k) {
            run += row[k];
            row[k] = run;
        }
    }
    __syncthreads();
    return all;
}

// One prefix difference per disc row; ACC is the accumulator type (float, int32, uint64).
template <int NOUT, typename T, typename ACC>
__device__ __forceinline__ void gather(const DiscArgs& p, const uint32_t* L, int stride, int col,
                                       int row_first, ACC (&acc)[NOUT]) {
#pragma unroll
    for (int k = 0; k < NOUT; ++k) acc[k] = (ACC)0;
    for (int d = 0; d < p.n_disc_rows; ++d) {
        const int packed = p.runs[d];  // wave-uniform: scalar load
        const int lo = packed & 0xffff;
        const int hi = packed >> 16;
        const T* base = reinterpret_cast<const T*>(L) + (row_first + d) * stride + col;
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            const T* rowp = base + (2 * k) * stride;
            acc[k] += (ACC)(T)(rowp[hi + 1] - rowp[lo]);
        }
    }
}

// value of one staged sample recovered from the prefix row: P[k+1] - P[k]
template <typename T>
__device__ __forceinline__ T sample(const uint32_t* L, int idx) {
    const T* q = reinterpret_cast<const T*>(L) + idx;
    return (T)(q[1] - q[0]);
}

template <bool WANT_TPI, bool WANT_STD, int TILE_H>
__global__ __launch_bounds__(kThreads) void disc_prefix_kernel(DiscArgs p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t L[];
    constexpr int NOUT = TILE_H / 2;  // 256 threads = 128 columns x 2 row phases

    const int cols_v = kTileW + p.halo_cols;       // staged values per LDS row
    const int stride = (cols_v + 1) | 1;           // +1 for the leading zero; odd: no bank clash
    const int rows_l = TILE_H + p.n_disc_rows - 1;

    const int ox0 = blockIdx.x * kTileW;
    // row tiles sit on GLOBAL multiples of TILE_H: a pixel is computed by the same instruction
    // sequence whatever row block it is part of
    const int oy0 = (p.out_row0 / TILE_H + (int)blockIdx.y) * TILE_H;
    const int gy0 = oy0 + p.dj_min;
    const int gx0 = ox0 + p.di_min;

    // integer offset near the local elevation (tile centre, clamped into the DEM)
    int cy = min(max(oy0 + TILE_H / 2, 0), p.gny - 1);
    cy = min(max(cy, p.in_row0), p.in_row0 + p.in_rows - 1);
    const int cx = min(ox0 + kTileW / 2, p.nx - 1);
    float c = truncf(p.in[(size_t)(cy - p.in_row0) * p.nx + cx]);
    if (!(fabsf(c) < 1.0e9f)) c = 0.0f;  // NaN/inf/huge centre: no offset
    const int ci = (int)c;

    const int col = threadIdx.x & (kTileW - 1);
    const int phase = threadIdx.x >> 7;
    // indices (without the per-output row term) of the pixel itself and of the zeroed tap
    const int self_idx = (phase - p.dj_min) * stride + col - p.di_min;
    const int ctr_idx = (phase + p.centre_dj - p.dj_min) * stride + col + p.centre_di - p.di_min;

    // per-output sums in float64 at the end: s1 = sum (trunc(x) - c), s2 = sum (trunc(x) - c)^2,
    // sf = sum frac(x); self_a / ctr_a = trunc(x) - c and self_f / ctr_f = frac(x) of the pixel
    // and of the zeroed tap
    double s1[NOUT], s2[NOUT], sf[NOUT], self_a[NOUT], ctr_a[NOUT], self_f[NOUT], ctr_f[NOUT];
#pragma unroll
    for (int k = 0; k < NOUT; ++k) s1[k] = s2[k] = sf[k] = self_a[k] = ctr_a[k] = self_f[k] = ctr_f[k] = 0.0;

    // Exact pipeline first: integer prefix sums cannot round.  ulim keeps one row-window sum
    // of u^2 below 2^32 so the wrap-around uint32 prefix differences stay exact.  Tiles with
    // non-finite or out-of-range samples take the float32 pipeline instead (NaN propagates).
    const float ulim = fminf(46000.0f, floorf(sqrtf(4294967295.0f / (float)(p.halo_cols + 1))));
    int flags = stage_and_scan<kPassU, int>(p, L, stride, rows_l, cols_v, gy0, gx0, c, ci, ulim);
    bool use_float = (flags & kFlagBad) != 0;
    if (!use_float) {
        int su[NOUT];
        gather<NOUT, int, int>(p, L, stride, col, phase, su);
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            s1[k] = (double)su[k];
            if (WANT_TPI) {
                self_a[k] = (double)sample<int>(L, self_idx + 2 * k * stride);
                ctr_a[k] = (double)sample<int>(L, ctr_idx + 2 * k * stride);
            }
        }
        if (WANT_STD) {
            __syncthreads();
            stage_and_scan<kPassU2, uint32_t>(p, L, stride, rows_l, cols_v, gy0, gx0, c, ci, ulim);
            unsigned long long q[NOUT];
            gather<NOUT, uint32_t, unsigned long long>(p, L, stride, col, phase, q);
#pragma unroll
            for (int k = 0; k < NOUT; ++k) s2[k] = (double)q[k];
        }
    }
    __syncthreads();
    if (use_float) {
        float fa[NOUT];
        stage_and_scan<kPassA, float>(p, L, stride, rows_l, cols_v, gy0, gx0, c, ci, 0.0f);
        gather<NOUT, float, float>(p, L, stride, col, phase, fa);
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            s1[k] = (double)fa[k];
            if (WANT_TPI) {
                self_a[k] = (double)sample<float>(L, self_idx + 2 * k * stride);
                ctr_a[k] = (double)sample<float>(L, ctr_idx + 2 * k * stride);
            }
        }
        if (WANT_STD) {
            __syncthreads();
            stage_and_scan<kPassT2, float>(p, L, stride, rows_l, cols_v, gy0, gx0, c, ci, 0.0f);
            gather<NOUT, float, float>(p, L, stride, col, phase, fa);
#pragma unroll
            for (int k = 0; k < NOUT; ++k) s2[k] = (double)fa[k];
            __syncthreads();
        }
    }
    if (flags & kFlagFrac) {
        // fractional parts: small positive floats, their float32 prefix sums are harmless
        float ff[NOUT];
        stage_and_scan<kPassF, float>(p, L, stride, rows_l, cols_v, gy0, gx0, c, ci, 0.0f);
        gather<NOUT, float, float>(p, L, stride, col, phase, ff);
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            sf[k] = (double)ff[k];
            if (use_float) s1[k] -= sf[k];  // keep s1 = sum of (trunc(x) - c) in both pipelines
            if (WANT_TPI) {
                self_f[k] = (double)sample<float>(L, self_idx + 2 * k * stride);
                ctr_f[k] = (double)sample<float>(L, ctr_idx + 2 * k * stride);
                if (use_float) {
                    self_a[k] -= self_f[k];
                    ctr_a[k] -= ctr_f[k];
                }
            }
        }
    }

    const int gx = ox0 + col;
    if (gx >= p.nx) return;
    const double n = (double)p.taps;
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
        const int oy = oy0 + phase + 2 * k;
        if (oy < p.out_row0 || oy >= p.out_row0 + p.out_rows) continue;
        const size_t o = (size_t)(oy - p.out_row0) * p.nx + gx;
        // With the padded taps staged as trunc = 0 (u = -c), s1 + n c is the exact integer sum of
        // trunc(x) over the in-domain taps: the results do not depend on the tile's choice of c.
        const double cd = (double)c;
        const double sum_x = (s1[k] + cd * n) + sf[k];
        if (WANT_TPI) {
            const double x_self = (self_a[k] + cd) + self_f[k];
            const double x_ctr = (ctr_a[k] + cd) + ctr_f[k];
            // (n-1) may be 0 for size 1 -> non-finite like the reference
            p.tpi[o] = (float)(x_self - (sum_x - x_ctr) / (n - 1.0));
        }
        if (WANT_STD) {
            const double sum_t2 = s2[k] + 2.0 * cd * s1[k] + cd * cd * n;
            double var = (sum_t2 - sum_x * sum_x / n) / (n - 1.0);
            if (var < 0.0) var = 0.0;  // keeps NaN, like np.clip
            p.sd[o] = (float)sqrt(var);
        }
    }
}

template <int TILE_H>
int launch_tile(const DiscArgs& a, dim3 grid, size_t lds, hipStream_t s, bool tpi, bool sd) {
    if (tpi && sd) {
        TOPO_HIP(hipFuncSetAttribute((const void*)disc_prefix_kernel<true, true, TILE_H>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((disc_prefix_kernel<true, true, TILE_H>), grid, dim3(kThreads), lds, s, a);
    } else if (tpi) {
        TOPO_HIP(hipFuncSetAttribute((const void*)disc_prefix_kernel<true, false, TILE_H>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((disc_prefix_kernel<true, false, TILE_H>), grid, dim3(kThreads), lds, s, a);
    } else {
        TOPO_HIP(hipFuncSetAttribute((const void*)disc_prefix_kernel<false, true, TILE_H>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((disc_prefix_kernel<false, true, TILE_H>), grid, dim3(kThreads), lds, s, a);
    }
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

}  // namespace

int build_disc(int size, DiscRuns* out) {
    TOPO_REQUIRE(size >= 1 && size <= 4096, "disc size %d out of range [1, 4096]", size);
    DiscRuns d;
    d.size = size;
    const int m = size / 2;         // int(size / 2), topo.py:205
    const int c = (size - 1) / 2;   // first kept index of mode="same"
    d.dj_min = c - (size - 1);
    d.dj_max = c;
    d.centre_dj = c - m;
    d.centre_di = c - m;
    d.lo.assign(size, 0);
    d.hi.assign(size, -1);
    d.di_min = 0;
    d.di_max = 0;
    bool first = true;
    for (int a = 0; a < size; ++a) {
        // kernel row a -> offset dj = c - a; its set columns b form one run
        int b_lo = size, b_hi = -1;
        for (int b = 0; b < size; ++b) {
            const bool on = (size < 5) ||
                            ((long)(a - m) * (a - m) + (long)(b - m) * (b - m) <= (long)m * m);
            if (on) {
                b_lo = b < b_lo ? b : b_lo;
                b_hi = b > b_hi ? b : b_hi;
                ++d.taps;
            }
        }
        const int row = (c - a) - d.dj_min;
        if (b_hi >= 0) {
            // di = c - b: the run [b_lo, b_hi] maps to [c - b_hi, c - b_lo]
            d.lo[row] = (int16_t)(c - b_hi);
            d.hi[row] = (int16_t)(c - b_lo);
            if (first || d.lo[row] < d.di_min) d.di_min = d.lo[row];
            if (first || d.hi[row] > d.di_max) d.di_max = d.hi[row];
            first = false;
        } else {
            d.lo[row] = 1;  // empty run: hi + 1 == lo
            d.hi[row] = 0;
        }
    }
    // trim empty rows at both ends (even sizes have none, kept for safety)
    *out = d;
    return TOPO_AMD_OK;
}

namespace {

// rows [r0, r0 + rows) of a plane of pitch src_pitch -> a plane of pitch dst_pitch; columns at and
// beyond `width` of the destination (dst_pitch > width) are set to 0
// (four columns per thread, 16-byte accesses wherever all four exist - they need dword alignment only, which the
// odd pitch has -: the one-column version copied at 3.3 TB/s)
__global__ __launch_bounds__(kThreads) void repitch_kernel(const float* src, int src_pitch, float* dst,
                                                           int dst_pitch, int width, int cols) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int x = (blockIdx.x * kThreads + threadIdx.x) * 4;
    if (x >= cols) return;
    const size_t r = blockIdx.y;
    const float* s = src + r * src_pitch + x;
    float* d = dst + r * dst_pitch + x;
    f4 v;
    if (x + 4 <= width) {
        v = *reinterpret_cast<const f4*>(s);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = x + e < width ? s[e] : 0.0f;
    }
    if (x + 4 <= cols) {
        *reinterpret_cast<f4*>(d) = v;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (x + e < cols) d[e] = v[e];
    }
}

int repitch(const float* src, int src_pitch, float* dst, int dst_pitch, int width, int cols, int rows) {
    if (rows <= 0) return TOPO_AMD_OK;
    TOPO_TRY(check_grid_rows(rows, "disc (re-pitched copy)"));
    dim3 grid(((cols + 3) / 4 + kThreads - 1) / kThreads, rows);
    hipLaunchKernelGGL(repitch_kernel, grid, dim3(kThreads), 0, ctx().compute, src, src_pitch, dst, dst_pitch,
                       width, cols);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

// The wave-shift kernels move 16-byte pieces of rows, so they need nx % 4 == 0 and 16-byte aligned
// planes; three DEM widths in four are not like that.  Such a block is copied into a plane whose
// pitch is nx rounded up to 4 with zeros in the extra columns, the kernels run on that plane as if
// it were the DEM, and the nx valid columns of the results are copied back.  Columns of zeros to
// the right of the DEM are exactly what the zero padding of the convolution stands for (n is the
// full tap count everywhere, see above), so every valid output is unchanged; the two copies cost
// about as much as a 7-pixel TPI, a fifth of what the generic kernel costs at 67 pixels.
int launch_disc_wave_repitched(const Block& b, int size, float* tpi_out, float* std_out) {
    const int nxp = (b.nx + 3) & ~3;
    void *in_p = nullptr, *tpi_p = nullptr, *std_p = nullptr;
    TOPO_TRY(workspace(4, (size_t)b.in_rows * nxp * sizeof(float), &in_p));
    if (tpi_out) TOPO_TRY(workspace(5, (size_t)b.out_rows * nxp * sizeof(float), &tpi_p));
    if (std_out) TOPO_TRY(workspace(6, (size_t)b.out_rows * nxp * sizeof(float), &std_p));
    TOPO_TRY(repitch(b.in, b.nx, (float*)in_p, nxp, b.nx, nxp, b.in_rows));
    Block bp = b;
    bp.in = (const float*)in_p;
    bp.nx = nxp;
    const int r = launch_disc_wave(bp, size, (float*)tpi_p, (float*)std_p);
    if (r != TOPO_AMD_OK) return r;
    if (tpi_out) TOPO_TRY(repitch((const float*)tpi_p, nxp, tpi_out, b.nx, b.nx, b.nx, b.out_rows));
    if (std_out) TOPO_TRY(repitch((const float*)std_p, nxp, std_out, b.nx, b.nx, b.nx, b.out_rows));
    return TOPO_AMD_OK;
}

}  // namespace

int launch_tpi_std(const Block& b, const DiscRuns& disc, float* tpi_out, float* std_out) {
    if (b.out_rows > kMaxLaunchRows) {
        ctx().seams.n = 0;  // (a sharded call this tall keeps its seams as launches of their own)
        for (int r = 0; r < b.out_rows; r += kMaxLaunchRows) {
            Block s = b;
            s.out_row0 = b.out_row0 + r;
            s.out_rows = std::min(kMaxLaunchRows, b.out_rows - r);
            TOPO_TRY(launch_tpi_std(s, disc, tpi_out ? tpi_out + (size_t)r * b.nx : nullptr, std_out ? std_out + (size_t)r * b.nx : nullptr));
        }
        return TOPO_AMD_OK;
    }
    TOPO_REQUIRE(tpi_out || std_out, "tpi_std: both outputs are NULL");
    Context& c = ctx();
    // the seam parts of a sharded call ride along only in the launchers of disc_wave_impl.hpp, on the block as it is
    // (not on a re-pitched copy); otherwise they stay with the caller, who sees the gate still armed
    if (c.seams.n > 0 && !(disc_wave_covers(disc.size) && b.nx % 4 == 0 && (reinterpret_cast<uintptr_t>(b.in) & 15) == 0 &&
                           (reinterpret_cast<uintptr_t>(std_out) & 15) == 0 && (reinterpret_cast<uintptr_t>(tpi_out) & 15) == 0))
        c.seams.n = 0;
    {
        int r = launch_disc_wave(b, disc.size, tpi_out, std_out);
        if (r == TOPO_AMD_EUNSUP && disc_wave_covers(disc.size) && b.nx >= 4)
            r = launch_disc_wave_repitched(b, disc.size, tpi_out, std_out);
        if (r != TOPO_AMD_EUNSUP) return r;
    }
    // sizes the wave-shift kernels do not cover (even, 1, 2, beyond 101): the prefix-plane path from
    // this size on, the LDS-gather kernel below it
    static const int big_from = [] {
        const char* e = std::getenv("TOPO_AMD_DISC_BIG_MIN_SIZE");
        // measured at 8192^2 (tools/generic_vs_big.py): size 66 TPI 2.24 ms (gather) / 2.05 ms (planes), STD
        // 4.23 / 4.44; size 84 4.79 / 2.40 and 9.51 / 5.41
        return e && *e ? std::atoi(e) : 70;
    }();
    if (disc.size >= big_from) return launch_disc_big(b, disc, tpi_out, std_out);
    const int n_rows = disc.dj_max - disc.dj_min + 1;
    const int halo_cols = disc.di_max - disc.di_min;
    const int cols_v = kTileW + halo_cols;
    const int stride = (cols_v + 1) | 1;
    int tile_h = 32;
    auto lds_for = [&](int th) { return (size_t)(th + n_rows - 1) * stride * sizeof(float); };
    const size_t lds_cap = 160 * 1024;
    while (tile_h > 8 && lds_for(tile_h) > lds_cap) tile_h /= 2;
    if (lds_for(tile_h) > lds_cap) return launch_disc_big(b, disc, tpi_out, std_out);
    // (32 rows are kept even where 16 would let two blocks share a CU: fewer halo re-reads beat
    // occupancy for this LDS-bound kernel)

    std::vector<int> packed(n_rows);
    for (int r = 0; r < n_rows; ++r) {
        const int lo = disc.lo[r] - disc.di_min;
        const int hi = disc.hi[r] - disc.di_min;  // empty run encodes hi + 1 == lo
        packed[r] = (lo & 0xffff) | (hi << 16);
    }
    void* d_runs = nullptr;
    TOPO_TRY(upload_table(0, packed.data(), packed.size() * sizeof(int), &d_runs));

    DiscArgs a;
    a.in = b.in;
    a.tpi = tpi_out;
    a.sd = std_out;
    a.runs = (const int*)d_runs;
    a.in_rows = b.in_rows;
    a.in_row0 = b.in_row0;
    a.gny = b.gny;
    a.nx = b.nx;
    a.out_row0 = b.out_row0;
    a.out_rows = b.out_rows;
    a.tile_h = tile_h;
    a.n_disc_rows = n_rows;
    a.dj_min = disc.dj_min;
    a.di_min = disc.di_min;
    a.halo_cols = halo_cols;
    a.centre_dj = disc.centre_dj;
    a.centre_di = disc.centre_di;
    a.taps = disc.taps;

    dim3 grid((b.nx + kTileW - 1) / kTileW,
              (b.out_row0 + b.out_rows - 1) / tile_h - b.out_row0 / tile_h + 1);
    const size_t lds = lds_for(tile_h);
    const bool tpi = tpi_out != nullptr, sd = std_out != nullptr;
    switch (tile_h) {
        case 32: return launch_tile<32>(a, grid, lds, c.compute, tpi, sd);
        case 16: return launch_tile<16>(a, grid, lds, c.compute, tpi, sd);
        default: return launch_tile<8>(a, grid, lds, c.compute, tpi, sd);
    }
}

}  // namespace topo
