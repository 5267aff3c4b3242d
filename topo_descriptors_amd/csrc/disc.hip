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

enum Pass { kPassU = 1, kPassU2 = 2, kPassF = 3, kPassInd = 4, kPassUL = 5, kPassU2L = 6 };
enum Flags { kFlagFrac = 1, kFlagBad = 2 };
constexpr float kOrdinaryLim = 262144.0f;   // 2^18 (kAbsLim of the wave-shift kernels)
constexpr float kMissingLim = 16777216.0f;  // 2^24: beyond it (or not finite) a sample is missing

__device__ __forceinline__ float load_padded(const DiscArgs& p, int gy, int gx) {
    // zero padding outside the global DEM; rows outside the block only feed unused outputs
    const int by = gy - p.in_row0;
    if (gy < 0 || gy >= p.gny || gx < 0 || gx >= p.nx || by < 0 || by >= p.in_rows) return 0.0f;
    return p.in[(size_t)by * p.nx + gx];
}

// What one staging pass writes to LDS for a DEM sample v (c is an integer-valued offset), always an integer:
//   U    int32  trunc(v) - c                U2   uint32 (trunc(v) - c)^2
//   F    int32  the fractional part in units of 2^-16 m (like the wave-shift kernels)
// and on tiles the 32-bit sums cannot hold (samples beyond 2^18, more relief than the row sums of u^2 take, non-finite
// samples), LIMBS that can (disc_wave_impl.hpp, Stage):
//   Ind  1 per sample that is not ordinary, + 2^16 per missing one        UL   u in two limbs (missing samples: u = 0)
//   U2L  u^2 in limbs of 16 bits, the last one taking what is left
template <int PASS>
__device__ __forceinline__ uint32_t transform(float v, int ci, int aux) {
    const float t = truncf(v);
    if (PASS == kPassF) return (uint32_t)(int)rintf((v - t) * 65536.0f);
    if (PASS == kPassInd || PASS == kPassUL || PASS == kPassU2L) {
        const bool missing = !(fabsf(t) < kMissingLim);
        if (PASS == kPassInd) return (missing ? 65536u : 0u) | ((missing || fabsf(t) > kOrdinaryLim) ? 1u : 0u);
        const int u = missing ? 0 : (int)t - ci;
        if (PASS == kPassUL) return aux == 0 ? ((uint32_t)u & 0xffffu) : (uint32_t)(u >> 16);
        const uint64_t w = (uint64_t)((int64_t)u * (int64_t)u) >> (16 * (aux & 3));
        return (aux & 4) ? (uint32_t)w : ((uint32_t)w & 0xffffu);
    }
    const int u = (int)t - ci;
    if (PASS == kPassU) return (uint32_t)u;
    return (uint32_t)u * (uint32_t)u;
}

// Stage one transformed tile and turn every LDS row into a prefix sum in place (uint32, modulo 2^32: differences are
// exact while a row-window sum stays below 2^32).  L[r][0] = 0, L[r][k] = sum of the first k staged values of row r.
// Returns block-wide flags (kPassU): fractional elevations present / samples the plain 32-bit passes cannot take
// (non-finite, |trunc(v)| > 2^18, or |trunc(v) - c| > ulim).
template <int PASS>
__device__ int stage_and_scan(const DiscArgs& p, uint32_t* L, int stride, int rows_l, int cols_v,
                              int gy0, int gx0, float c, int ci, float ulim, int aux = 0) {
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
                if (!(fabsf(t - c) <= ulim) || !(fabsf(t) <= kOrdinaryLim)) flags |= kFlagBad;  // also catches NaN / inf
            }
            row[k] = transform<PASS>(v, ci, aux);
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
        uint32_t* row = L + r * stride;
        uint32_t run = 0;
        row[0] = 0;
        int k = 1;
        for (; k + 3 <= cols_v; k += 4) {
            const uint32_t v0 = row[k], v1 = row[k + 1], v2 = row[k + 2], v3 = row[k + 3];
            const uint32_t s0 = run + v0;
            const uint32_t s1 = s0 + v1;
            const uint32_t s2 = s1 + v2;
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

// One prefix difference per disc row (exact modulo 2^32), accumulated in 64 bits; SIGNED: the differences are int32.
template <int NOUT, bool SIGNED>
__device__ __forceinline__ void gather(const DiscArgs& p, const uint32_t* L, int stride, int col, int row_first, int64_t (&acc)[NOUT]) {
#pragma unroll
    for (int k = 0; k < NOUT; ++k) acc[k] = 0;
    for (int d = 0; d < p.n_disc_rows; ++d) {
        const int packed = p.runs[d];  // wave-uniform: scalar load
        const int lo = packed & 0xffff;
        const int hi = packed >> 16;
        const uint32_t* base = L + (row_first + d) * stride + col;
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            const uint32_t* rowp = base + (2 * k) * stride;
            const uint32_t dlt = rowp[hi + 1] - rowp[lo];
            acc[k] += SIGNED ? (int64_t)(int32_t)dlt : (int64_t)(uint64_t)dlt;
        }
    }
}

__device__ __forceinline__ double int128_to_double(__int128 v) {
    const bool neg = v < 0;
    const unsigned __int128 mag = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
    const double d = (double)(uint64_t)(mag >> 64) * 18446744073709551616.0 + (double)(uint64_t)mag;
    return neg ? -d : d;
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

    // integer offset near the local elevation (tile centre, clamped into the DEM and the block's view).  Every sum below is
    // an exact integer and the finalisation removes c again, so the results do not depend on it.
    int cy = min(max(oy0 + TILE_H / 2, 0), p.gny - 1);
    cy = min(max(cy, p.in_row0), p.in_row0 + p.in_rows - 1);
    const int cx = min(ox0 + kTileW / 2, p.nx - 1);
    float c = truncf(p.in[(size_t)(cy - p.in_row0) * p.nx + cx]);
    if (!(fabsf(c) < kMissingLim)) c = 0.0f;  // NaN/inf/huge centre: no offset
    const int ci = (int)c;

    const int col = threadIdx.x & (kTileW - 1);
    const int phase = threadIdx.x >> 7;

    // exact per-output sums over the n taps (a tap outside the DEM reads x = 0, the zero padding of mode="same")
    int64_t su[NOUT], su2[NOUT], sg[NOUT], ind[NOUT];
#pragma unroll
    for (int k = 0; k < NOUT; ++k) su[k] = su2[k] = sg[k] = ind[k] = 0;

    // ulim keeps one row-window sum of u^2 below 2^32 so that the wrap-around uint32 prefix differences stay exact
    const float ulim = fminf(46000.0f, floorf(sqrtf(4294967295.0f / (float)(p.halo_cols + 1))));
    const int flags = stage_and_scan<kPassU>(p, L, stride, rows_l, cols_v, gy0, gx0, c, ci, ulim);
    const bool limbs = (flags & kFlagBad) != 0;
    if (!limbs) {
        gather<NOUT, true>(p, L, stride, col, phase, su);
        if (WANT_STD) {
            __syncthreads();
            stage_and_scan<kPassU2>(p, L, stride, rows_l, cols_v, gy0, gx0, c, ci, ulim);
            gather<NOUT, false>(p, L, stride, col, phase, su2);
        }
    } else {
        int64_t part[NOUT];
        __syncthreads();
        stage_and_scan<kPassInd>(p, L, stride, rows_l, cols_v, gy0, gx0, c, ci, ulim);
        gather<NOUT, false>(p, L, stride, col, phase, ind);
        __syncthreads();
        stage_and_scan<kPassUL>(p, L, stride, rows_l, cols_v, gy0, gx0, c, ci, ulim, 0);
        gather<NOUT, false>(p, L, stride, col, phase, su);
        __syncthreads();
        stage_and_scan<kPassUL>(p, L, stride, rows_l, cols_v, gy0, gx0, c, ci, ulim, 1);
        gather<NOUT, true>(p, L, stride, col, phase, part);
#pragma unroll
        for (int k = 0; k < NOUT; ++k) su[k] += part[k] * 65536;
        if (WANT_STD) {
            for (int limb = 0; limb < 3; ++limb) {
                __syncthreads();
                stage_and_scan<kPassU2L>(p, L, stride, rows_l, cols_v, gy0, gx0, c, ci, ulim, limb | (limb == 2 ? 4 : 0));
                gather<NOUT, false>(p, L, stride, col, phase, part);
#pragma unroll
                for (int k = 0; k < NOUT; ++k) su2[k] += (int64_t)((uint64_t)part[k] << (16 * limb));
            }
        }
    }
    if (flags & kFlagFrac) {
        __syncthreads();
        stage_and_scan<kPassF>(p, L, stride, rows_l, cols_v, gy0, gx0, c, ci, ulim);
        gather<NOUT, true>(p, L, stride, col, phase, sg);
    }

    const int gx = ox0 + col;
    if (gx >= p.nx) return;
    const double n = (double)p.taps;
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
        const int oy = oy0 + phase + 2 * k;
        if (oy < p.out_row0 || oy >= p.out_row0 + p.out_rows) continue;
        const size_t o = (size_t)(oy - p.out_row0) * p.nx + gx;
        // load_padded gives x = 0 for a tap outside the DEM, staged as u = -c like any other sample: with the full tap count n,
        // sum trunc(x) = Su + c n and sum trunc(x)^2 = Su2 + 2 c Su + c^2 n hold as they stand (exact integers: c drops out).
        const __int128 cI = ci;
        const int64_t T = su[k] + (int64_t)ci * (int64_t)p.taps;              // sum of trunc(x) over the n taps (padding: 0)
        const __int128 S2 = (__int128)su2[k] + 2 * cI * su[k] + cI * cI * p.taps;  // sum of trunc(x)^2
        const double sf = (double)sg[k] * (1.0 / 65536.0);
        const double sum_x = (double)T + sf;
        const bool missing = (ind[k] >> 16) != 0, ordinary = (ind[k] & 0xffff) == 0;
        if (WANT_TPI) {
            const int sy = oy, sx = gx;
            const int cy2 = oy + p.centre_dj, cx2 = gx + p.centre_di;
            const double x_self = (double)load_padded(p, sy, sx);
            const double x_ctr = (double)load_padded(p, cy2, cx2);
            // (n-1) may be 0 for size 1 -> non-finite like the reference
            const float v = (float)(x_self - (sum_x - x_ctr) / (n - 1.0));
            p.tpi[o] = missing ? __uint_as_float(0x7fc00000u) : v;
        }
        if (WANT_STD) {
            float v;
            if (ordinary) {
                const double sum_t2 = (double)(int64_t)S2;  // below 2^53
                double var = (sum_t2 - sum_x * sum_x / n) / (n - 1.0);
                if (var < 0.0) var = 0.0;  // keeps NaN, like np.clip
                v = (float)sqrt(var);
            } else {
                // samples beyond 2^18 in the disc: 2^32 (n s2 - s1^2) exactly in 128 bits (disc_wave_impl.hpp, std_from_exact_sums)
                const __int128 A = (__int128)p.taps * S2 - (__int128)T * T;
                const __int128 num = (A << 32) - (((__int128)T * sg[k]) << 17) - (__int128)sg[k] * sg[k];
                double var = int128_to_double(num) * (1.0 / 4294967296.0) / (n * (n - 1.0));
                if (var < 0.0) var = 0.0;
                v = (float)sqrt(var);
            }
            p.sd[o] = missing ? __uint_as_float(0x7fc00000u) : v;
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
    constexpr int big_from = 70;
    // measured at 8192^2 (tools/generic_vs_big.py): size 66 TPI 2.24 ms (gather) / 2.05 ms (planes), STD
    // 4.23 / 4.44; size 84 4.79 / 2.40 and 9.51 / 5.41
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
