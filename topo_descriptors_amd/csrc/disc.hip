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

enum Pass { kPassA = 0, kPassU2 = 1, kPassF = 2 };

__device__ __forceinline__ float load_padded(const DiscArgs& p, int gy, int gx) {
    // zero padding outside the global DEM; rows outside the block only feed unused outputs
    const int by = gy - p.in_row0;
    if (gy < 0 || gy >= p.gny || gx < 0 || gx >= p.nx || by < 0 || by >= p.in_rows) return 0.0f;
    return p.in[(size_t)by * p.nx + gx];
}

template <int PASS>
__device__ __forceinline__ float transform(float v, float c) {
    if (PASS == kPassA) return v - c;
    const float t = truncf(v);
    if (PASS == kPassU2) {
        const float u = t - c;
        return u * u;
    }
    return v - t;
}

// Stage one transformed tile and turn every LDS row into an exclusive-start prefix sum:
// L[r][0] = 0, L[r][k] = sum of the first k staged values of row r.
// Returns (block-wide) whether any staged elevation had a fractional part.
template <int PASS>
__device__ bool stage_and_scan(const DiscArgs& p, float* L, int stride, int rows_l, int cols_v,
                               int gy0, int gx0, float c) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    bool frac = false;
    for (int r = wave; r < rows_l; r += kThreads / 64) {
        const int gy = gy0 + r;
        float* row = L + r * stride + 1;
        for (int k = lane; k < cols_v; k += 64) {
            const float v = load_padded(p, gy, gx0 + k);
            if (PASS == kPassU2) frac |= (v != truncf(v));
            row[k] = transform<PASS>(v, c);
        }
    }
    const bool any_frac = __syncthreads_or(frac);
    for (int r = threadIdx.x; r < rows_l; r += kThreads) {
        float* row = L + r * stride;
        float run = 0.0f;
        row[0] = 0.0f;
        int k = 1;
        for (; k + 3 <= cols_v; k += 4) {
            const float v0 = row[k], v1 = row[k + 1], v2 = row[k + 2], v3 = row[k + 3];
            const float s0 = run + v0;
            const float s1 = s0 + v1;
            const float s2 = s1 + v2;
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
    return any_frac;
}

template <int NOUT>
__device__ __forceinline__ void gather(const DiscArgs& p, const float* L, int stride, int col,
                                       int row_first, float (&acc)[NOUT]) {
#pragma unroll
    for (int k = 0; k < NOUT; ++k) acc[k] = 0.0f;
    for (int d = 0; d < p.n_disc_rows; ++d) {
        const int packed = p.runs[d];  // wave-uniform: scalar load
        const int lo = packed & 0xffff;
        const int hi = packed >> 16;
        const float* base = L + (row_first + d) * stride + col;
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            const float* rowp = base + (2 * k) * stride;
            acc[k] += rowp[hi + 1] - rowp[lo];
        }
    }
}

template <bool WANT_TPI, bool WANT_STD, int TILE_H>
__global__ __launch_bounds__(kThreads) void disc_prefix_kernel(DiscArgs p) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    constexpr int NOUT = TILE_H / 2;  // 256 threads = 128 columns x 2 row phases

    const int cols_v = kTileW + p.halo_cols;       // staged values per LDS row
    const int stride = (cols_v + 1) | 1;           // +1 for the leading zero; odd: no bank clash
    const int rows_l = TILE_H + p.n_disc_rows - 1;

    const int ox0 = blockIdx.x * kTileW;
    const int oy0 = p.out_row0 + blockIdx.y * TILE_H;
    const int gy0 = oy0 + p.dj_min;
    const int gx0 = ox0 + p.di_min;

    // integer offset near the local elevation (tile centre, clamped into the DEM)
    int cy = min(max(oy0 + TILE_H / 2, 0), p.gny - 1);
    cy = min(max(cy, p.in_row0), p.in_row0 + p.in_rows - 1);
    const int cx = min(ox0 + kTileW / 2, p.nx - 1);
    float c = truncf(p.in[(size_t)(cy - p.in_row0) * p.nx + cx]);
    if (!(fabsf(c) < 1e30f)) c = 0.0f;  // NaN/inf centre: fall back to no offset

    const int col = threadIdx.x & (kTileW - 1);
    const int phase = threadIdx.x >> 7;

    float sum_a[NOUT], sum_u2[NOUT], sum_f[NOUT];
    float self_a[NOUT], ctr_a[NOUT];

    stage_and_scan<kPassA>(p, L, stride, rows_l, cols_v, gy0, gx0, c);
    gather<NOUT>(p, L, stride, col, phase, sum_a);
    if (WANT_TPI) {
        // a(j,i) itself and the zeroed tap, recovered from the prefix rows
        const int self_r = -p.dj_min, self_c = col - p.di_min;
        const int ctr_r = p.centre_dj - p.dj_min, ctr_c = col + p.centre_di - p.di_min;
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            const float* rs = L + (phase + 2 * k + self_r) * stride + self_c;
            self_a[k] = rs[1] - rs[0];
            const float* rc = L + (phase + 2 * k + ctr_r) * stride + ctr_c;
            ctr_a[k] = rc[1] - rc[0];
        }
    }
    bool any_frac = false;
    if (WANT_STD) {
        __syncthreads();
        // the fractional-part sum only matters when the tile holds non-integer elevations
        any_frac = stage_and_scan<kPassU2>(p, L, stride, rows_l, cols_v, gy0, gx0, c);
        gather<NOUT>(p, L, stride, col, phase, sum_u2);
        if (any_frac) {
            __syncthreads();
            stage_and_scan<kPassF>(p, L, stride, rows_l, cols_v, gy0, gx0, c);
            gather<NOUT>(p, L, stride, col, phase, sum_f);
        }
    }

    const int gx = ox0 + col;
    if (gx >= p.nx) return;
    const double n = (double)p.taps;
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
        const int oy = oy0 + phase + 2 * k;
        if (oy >= p.out_row0 + p.out_rows) continue;
        const size_t o = (size_t)(oy - p.out_row0) * p.nx + gx;
        if (WANT_TPI) {
            // division kept in float64: (n-1) may be 0 for size 1 -> non-finite like the reference
            const double mean_excl = ((double)sum_a[k] - (double)ctr_a[k]) / (n - 1.0);
            p.tpi[o] = (float)((double)self_a[k] - mean_excl);
        }
        if (WANT_STD) {
            const double s1 = (double)sum_a[k];
            double num = (double)sum_u2[k] - s1 * s1 / n;
            if (any_frac) num -= 2.0 * (double)c * (double)sum_f[k];
            double var = num / (n - 1.0);
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

int launch_tpi_std(const Block& b, const DiscRuns& disc, float* tpi_out, float* std_out) {
    TOPO_REQUIRE(tpi_out || std_out, "tpi_std: both outputs are NULL");
    Context& c = ctx();
    const int n_rows = disc.dj_max - disc.dj_min + 1;
    const int halo_cols = disc.di_max - disc.di_min;
    const int cols_v = kTileW + halo_cols;
    const int stride = (cols_v + 1) | 1;
    int tile_h = 32;
    auto lds_for = [&](int th) { return (size_t)(th + n_rows - 1) * stride * sizeof(float); };
    const size_t lds_cap = 160 * 1024;
    while (tile_h > 8 && lds_for(tile_h) > lds_cap) tile_h /= 2;
    if (lds_for(tile_h) > lds_cap) {
        set_error("tpi/std: disc size %d needs %zu B of LDS per tile (limit %zu); "
                  "the large-radius path is not built yet", disc.size, lds_for(tile_h), lds_cap);
        return TOPO_AMD_EUNSUP;
    }
    // prefer two resident blocks per CU when the tile allows it
    if (tile_h == 32 && lds_for(32) > lds_cap / 2 && lds_for(16) <= lds_cap / 2) {
        // keep 32: fewer halo re-reads beats occupancy for this LDS-bound kernel
    }

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

    dim3 grid((b.nx + kTileW - 1) / kTileW, (b.out_rows + tile_h - 1) / tile_h);
    const size_t lds = lds_for(tile_h);
    const bool tpi = tpi_out != nullptr, sd = std_out != nullptr;
    switch (tile_h) {
        case 32: return launch_tile<32>(a, grid, lds, c.compute, tpi, sd);
        case 16: return launch_tile<16>(a, grid, lds, c.compute, tpi, sd);
        default: return launch_tile<8>(a, grid, lds, c.compute, tpi, sd);
    }
}

}  // namespace topo
