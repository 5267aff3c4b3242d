// K3/K4/K5: separable Gaussian, Sobel, and the gradient -> slope/aspect epilogue.
//
// Replaces scipy.ndimage.gaussian_filter (topo.py:80, :173, :298, :631-635), numpy.gradient
// plus the normalisation / slope / aspect arithmetic (topo.py:631-642, :707-712) and
// scipy.ndimage.convolve with the Sobel pair (topo.py:679-683).
//
// Gaussian = two 1-D passes like scipy (axis 0, then axis 1), radius int(4 sigma + 0.5),
// float64-normalised taps rounded to float32, reflect ("symmetric") boundary at the GLOBAL
// DEM edges, and the intermediate rounded to float32 between the passes exactly as the
// reference does.  Each thread register-blocks TB consecutive outputs along the filter
// axis so that one loaded sample feeds TB FMAs; samples are taken relative to a per-thread
// offset so float32 accumulation keeps its digits on 2000 m terrain.
#include "common.hpp"
#include "atan.hpp"

#include <cstdlib>

#include <cmath>

namespace topo {

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ int reflect_index(int i, int n) {
    // scipy mode="reflect": (d c b a | a b c d | d c b a), period 2n
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

// The accumulation offset of a group of outputs is one of its own samples; a non-finite sample cannot serve
// (x - NaN is NaN for every x of the group).
__device__ __forceinline__ float finite_or_zero(float c) { return fabsf(c) <= 3.0e38f ? c : 0.0f; }

struct GaussArgs {
    const float* in;
    float* out;
    const float* taps;  // 2R+1 taps followed by zeros up to a multiple of the tap chunk
    int in_rows, in_row0, gny, nx;
    int out_row0, out_rows;
    int radius, nchunks;
    int group0;  // axis 0: first row group (global numbering) that intersects the output rows
    unsigned char* flags;  // matrix-core kernels: one byte per 32 x 32 output tile, 1 = an output is not finite
    float tap_scale, out_scale;  // f16 matrix-core kernels: taps are multiplied by tap_scale (a power of two), sums by out_scale
    int* wild_flag;     // fused kernel: set to 1 when it stages a sample that is not a plain finite one
    uint32_t* wild_host;  // ... and this pinned host word of the DEM's memo entry (dem_memo_wild), or nullptr
    const int* run_if;  // two-pass f16 kernels behind a fused launch: return at once unless *run_if != 0 (nullptr: run)
    const float* wtab;  // split-once kernels: W[5][64], the taps an output lays over each slab of 64 (behind the plain taps)
    int fine_rows, fine_cols;  // split-once axis-1 kernel: flags is [fine_rows bands of 16 rows][fine_cols tiles of 32 columns] (0: one byte per 32 x 32 unit)
};

// Register tiling shared by both axes: a thread produces TB consecutive outputs along the
// filter axis, out[t] = sum_k w[k] in[t + k].  The taps are walked in chunks of KB; a chunk needs
// the TB + KB - 1 samples in[c KB ... c KB + TB + KB - 2], of which only KB are new, so per chunk
// the thread fetches KB samples (all in flight together, one chunk ahead of the FMAs) and issues
// TB x KB FMAs against KB wave-uniform taps (one scalar load per chunk).
// The padded zero taps of the last chunk are multiplied in like the others: skipping them (a second copy of
// the unrolled chunk body, bounded by the true tap count) was measured and costs 28 % of the gradient at
// sigma 3.25 (10.9 ms against 8.5 ms at 32768^2).  What it would buy: 0 x NaN = NaN, so a non-finite sample
// reaches up to KB - 1 outputs further along each axis (towards lower indices) than in ndimage.gaussian_filter;
// finite data is unaffected.  tests/test_gpu_parity.py::test_gaussian_nan_footprint pins that bound.
template <int TB, int KB, class Fetch>
__device__ __forceinline__ void tap_chunks(const float* taps, int nchunks, float c, Fetch fetch,
                                           float (&acc)[TB]) {
    float win[TB + KB - 1];
    float nxt[KB];
#pragma unroll
    for (int t = 0; t < TB; ++t) acc[t] = 0.0f;
#pragma unroll
    for (int i = 0; i < TB - 1; ++i) win[i] = fetch(i) - c;
#pragma unroll
    for (int i = 0; i < KB; ++i) nxt[i] = fetch(TB - 1 + i);
    for (int ch = 0; ch < nchunks; ++ch) {
#pragma unroll
        for (int i = 0; i < KB; ++i) win[TB - 1 + i] = nxt[i] - c;
        if (ch + 1 < nchunks) {
#pragma unroll
            for (int i = 0; i < KB; ++i) nxt[i] = fetch((ch + 1) * KB + TB - 1 + i);
        }
        const float* w = taps + ch * KB;  // wave-uniform
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
            const float wk = w[kk];
#pragma unroll
            for (int t = 0; t < TB; ++t) acc[t] = fmaf(wk, win[t + kk], acc[t]);
        }
#pragma unroll
        for (int i = 0; i < TB - 1; ++i) win[i] = win[i + KB];
    }
}

// ---- axis 0 (down the columns): lanes run along x, plain coalesced global loads ------------
// Row groups are aligned to GLOBAL multiples of TB and the accumulation offset c is a sample of
// the group itself, so an output pixel is computed by exactly the same instruction sequence
// whatever row block it is part of (shards stay bit-identical to the single-block run).
template <int TB, int KB>
__global__ __launch_bounds__(kThreads) void gauss_axis0_kernel(GaussArgs p) {
    const int x = blockIdx.x * kThreads + threadIdx.x;
    if (x >= p.nx) return;
    const int y0 = (p.group0 + (int)blockIdx.y) * TB;
    const int first = y0 - p.radius;                       // global row of sample 0
    const int last = first + TB - 1 + p.nchunks * KB - 1;  // last sample fetched (the ones past 2 radius + TB - 1 meet no tap)
    const bool plain = first >= 0 && last < p.gny && first >= p.in_row0 && last < p.in_row0 + p.in_rows;
    const float* col = p.in + x;
    auto fetch = [&](int i) -> float {
        int gy = first + i;
        if (!plain) {  // wave-uniform: reflect at the global edges, clamp into the block
            gy = reflect_index(gy, p.gny);
            gy = min(max(gy, p.in_row0), p.in_row0 + p.in_rows - 1);
        }
        return col[(size_t)(gy - p.in_row0) * p.nx];
    };
    // the middle row of the group lies within `radius` rows of every row of the group, hence
    // inside any block that computes part of it - provided radius >= TB/2 - 1; tiny filters
    // accumulate without an offset (they have nothing to lose)
    // (a NaN or inf there would make every output of the group non-finite instead of the ones its taps reach)
    const float c = finite_or_zero(p.radius >= TB / 2 - 1 ? fetch(p.radius + TB / 2) : 0.0f);
    float acc[TB];
    tap_chunks<TB, KB>(p.taps, p.nchunks, c, fetch, acc);
#pragma unroll
    for (int t = 0; t < TB; ++t) {
        const int oy = y0 + t;
        if (oy >= p.out_row0 && oy < p.out_row0 + p.out_rows)
            p.out[(size_t)(oy - p.out_row0) * p.nx + x] = c + acc[t];
    }
}

// ---- axis 1 (along the rows): tile staged in LDS, lanes run along y ------------------------
// block = 64 rows x (NW * TB) output columns; wave w owns TB consecutive columns of every row.
template <int TB, int KB, int NW>
__global__ __launch_bounds__(NW * 64) void gauss_axis1_kernel(GaussArgs p) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    constexpr int TR = 64;
    constexpr int TC = NW * TB;
    const int R = p.radius;
    const int cols_l = TC + p.nchunks * KB - 1;  // samples a row of the tile can be asked for
    const int stride = cols_l | 1;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ox0 = blockIdx.x * TC;
    const int oy0 = blockIdx.y * TR;  // relative to out_row0; rows here need no halo

    // stage: wave per row, lanes along x (coalesced); tiles that reach past the left/right DEM
    // edge reflect there, all others copy straight
    const bool inner = ox0 - R >= 0 && ox0 - R + cols_l <= p.nx;
    for (int r = wave; r < TR; r += NW) {
        const int row = min(oy0 + r, p.out_rows - 1);
        const float* src = p.in + (size_t)(row + p.out_row0 - p.in_row0) * p.nx;
        float* dst = L + r * stride;
        if (inner) {
            const float* s0 = src + (ox0 - R);
            for (int k = lane; k < cols_l; k += 64) dst[k] = s0[k];
        } else {
            for (int k = lane; k < cols_l; k += 64) dst[k] = src[reflect_index(ox0 - R + k, p.nx)];
        }
    }
    __syncthreads();

    const float* rowp = L + lane * stride + wave * TB;
    auto fetch = [&](int i) -> float { return rowp[i]; };
    const float c = finite_or_zero(rowp[R]);
    float acc[TB];
    tap_chunks<TB, KB>(p.taps, p.nchunks, c, fetch, acc);
    __syncthreads();
    // transpose back through LDS so the stores are row-coalesced
    constexpr int ostride = TC + 1;
    float* O = L;
#pragma unroll
    for (int t = 0; t < TB; ++t) O[lane * ostride + wave * TB + t] = c + acc[t];
    __syncthreads();
    for (int idx = threadIdx.x; idx < TR * TC; idx += NW * 64) {
        const int r = idx / TC, k = idx % TC;
        const int oy = oy0 + r, ox = ox0 + k;
        if (oy < p.out_rows && ox < p.nx) p.out[(size_t)oy * p.nx + ox] = O[r * ostride + k];
    }
}

// ---- gradient epilogue ---------------------------------------------------------------------
struct GradArgs {
    const float* gx_src;  // smoothed field differenced along x (rows start at s_row0)
    const float* gy_src;  // smoothed field differenced along y
    const float* raw;     // Sobel branch: the DEM block itself
    int raw_rows, raw_row0;
    int s_row0, s_rows;
    int gny, nx, out_row0, out_rows;
    int res_mode;
    const float* res_x;   // device: 1 value, nx values, or out_rows*nx values
    const float* res_y;   // device: 1 value, gny values, or out_rows*nx values
    float *dx, *dy, *slope, *aspect;
    const int* run_if;  // nullptr, or: return at once unless *run_if != 0 (the epilogue behind a deferred two-pass smooth)
};

// ---- slope / aspect arithmetic -----------------------------------------------------------------
// The libm routines (atanf, atan2f, IEEE division with scaling) cost ~300 instructions per pixel
// together and made the epilogue VALU-bound at 3 TB/s.  These replacements keep float32-level
// accuracy (atan: Abramowitz & Stegun 4.4.49, |err| <= 2e-8 on [0,1]; division: reciprocal plus
// one FMA correction, correctly rounded except in rare half-way cases) and the IEEE special cases
// the descriptors rely on: signed zeros (flat terrain: dy = -0 -> atan2(+0,-0) = pi -> aspect 0),
// NaN propagation, slope 90 for infinite gradients.
__device__ __forceinline__ float div_f32(float a, float b) {
    const float r = __builtin_amdgcn_rcpf(b);
    const float q = a * r;
    return fmaf(fmaf(-q, b, a), r, q);
}

__device__ __forceinline__ float atan2_f32(float y, float x) {
    const float a = fabsf(y), b = fabsf(x);
    const float mx = fmaxf(a, b), mn = fminf(a, b);
    const float t = mx == 0.0f ? 0.0f : mn * __builtin_amdgcn_rcpf(mx);
    float r = atan_unit(t);
    if (a > b) r = 1.5707963267948966f - r;
    if (__float_as_uint(x) >> 31) r = 3.14159265358979323846f - r;  // sign BIT: x = -0 counts
    r = copysignf(r, y);
    return (x != x || y != y) ? NAN : r;
}

__device__ __forceinline__ void resolution_at(const GradArgs& p, int oy, int ox, float& rx, float& ry) {
    if (p.res_mode == TOPO_AMD_RES_SCALAR) {
        rx = p.res_x[0];
        ry = p.res_y[0];
    } else if (p.res_mode == TOPO_AMD_RES_1D) {
        rx = p.res_x[ox];
        ry = p.res_y[oy];
    } else {
        const size_t o = (size_t)(oy - p.out_row0) * p.nx + ox;
        rx = p.res_x[o];
        ry = p.res_y[o];
    }
}

// dx, dy in -> normalised dx, dy, slope and aspect in degrees (topo.py:637-642)
__device__ __forceinline__ void gradient_values(float& dx, float& dy, float rx, float ry, float& slope,
                                                float& aspect) {
    dx = div_f32(dx, rx);  // signed resolutions: -0.0 must survive (aspect of flat terrain)
    dy = div_f32(dy, ry);
    const float rad2deg = 57.29577951308232f;
    const float d2 = dx * dx;
    const float e2 = dy * dy;
    slope = atan_pos(__builtin_amdgcn_sqrtf(d2 + e2)) * rad2deg;
    float a = 180.0f + atan2_f32(dx, dy) * rad2deg;
    if (a >= 360.0f) a -= 360.0f;  // float32 "% 360" of a value in [0, 360]
    aspect = a;
}

__device__ __forceinline__ void finish_gradient(const GradArgs& p, int oy, int ox, float dx,
                                                float dy) {
    const size_t o = (size_t)(oy - p.out_row0) * p.nx + ox;
    float rx, ry, slope, aspect;
    resolution_at(p, oy, ox, rx, ry);
    gradient_values(dx, dy, rx, ry, slope, aspect);
    if (p.dx) p.dx[o] = dx;
    if (p.dy) p.dy[o] = dy;
    if (p.slope) p.slope[o] = slope;
    if (p.aspect) p.aspect[o] = aspect;
}

__device__ __forceinline__ void epilogue_pixel(const GradArgs& p, int oy, int ox);
__global__ __launch_bounds__(kThreads) void gradient_epilogue_kernel(GradArgs p) {
    const int ox = blockIdx.x * kThreads + threadIdx.x;
    const int oy = p.out_row0 + blockIdx.y;
    if (ox >= p.nx) return;
    epilogue_pixel(p, oy, ox);
}
// the epilogue behind a deferred two-pass smooth: nothing unless *p.run_if != 0; a grid of a few hundred rows that
// strides over the output rows (a million blocks that only read the flag would take a few hundred microseconds)
__global__ __launch_bounds__(kThreads) void gradient_epilogue_if_kernel(GradArgs p) {
    if (*p.run_if == 0) return;
    const int ox = blockIdx.x * kThreads + threadIdx.x;
    if (ox >= p.nx) return;
    for (int oy = p.out_row0 + blockIdx.y; oy < p.out_row0 + p.out_rows; oy += gridDim.y) epilogue_pixel(p, oy, ox);
}
__device__ __forceinline__ void epilogue_pixel(const GradArgs& p, int oy, int ox) {
    // numpy.gradient: central difference / 2 inside, first-order one-sided at the edges
    const float* rowx = p.gx_src + (size_t)(oy - p.s_row0) * p.nx;
    float dx;
    if (ox == 0) dx = rowx[1] - rowx[0];
    else if (ox == p.nx - 1) dx = rowx[ox] - rowx[ox - 1];
    else dx = (rowx[ox + 1] - rowx[ox - 1]) * 0.5f;
    const float* coly = p.gy_src + (size_t)(oy - p.s_row0) * p.nx + ox;
    float dy;
    if (oy == 0) dy = coly[p.nx] - coly[0];
    else if (oy == p.gny - 1) dy = coly[0] - coly[-p.nx];
    else dy = (coly[p.nx] - coly[-p.nx]) * 0.5f;
    finish_gradient(p, oy, ox, dx, dy);
}

// The same, four adjacent pixels per thread: 16-byte loads of the three smoothed rows and 16-byte stores of
// the four outputs (the one-pixel kernel moves the 20 B/pixel as dwords and ran at 3.2 TB/s).  Identical arithmetic
// per pixel, hence identical bits.  Any width from 8 columns (round 3: the 16-byte accesses only need dword alignment;
// the last 1 ... 3 pixels of a row whose length is not a multiple of 4 go one by one).
__device__ __forceinline__ void epilogue_row4(const GradArgs& p, int oy, int ox);
__global__ __launch_bounds__(kThreads) void gradient_epilogue4_kernel(GradArgs p) {
    const int ox = (blockIdx.x * kThreads + threadIdx.x) * 4;
    const int oy = p.out_row0 + blockIdx.y;
    if (ox >= p.nx) return;
    epilogue_row4(p, oy, ox);
}
__global__ __launch_bounds__(kThreads) void gradient_epilogue4_if_kernel(GradArgs p) {  // (see gradient_epilogue_if_kernel)
    if (*p.run_if == 0) return;
    const int ox = (blockIdx.x * kThreads + threadIdx.x) * 4;
    if (ox >= p.nx) return;
    for (int oy = p.out_row0 + blockIdx.y; oy < p.out_row0 + p.out_rows; oy += gridDim.y) epilogue_row4(p, oy, ox);
}
__device__ __forceinline__ void epilogue_row4(const GradArgs& p, int oy, int ox) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (ox + 4 > p.nx) {
        for (int x = ox; x < p.nx; ++x) epilogue_pixel(p, oy, x);
        return;
    }
    const float* rowx = p.gx_src + (size_t)(oy - p.s_row0) * p.nx;
    const f4 mid = *reinterpret_cast<const f4*>(rowx + ox);
    const float left = ox > 0 ? rowx[ox - 1] : 0.0f, right = ox + 4 < p.nx ? rowx[ox + 4] : 0.0f;
    const float* coly = p.gy_src + (size_t)(oy - p.s_row0) * p.nx + ox;
    const f4 cen = *reinterpret_cast<const f4*>(coly);
    const f4 up = oy > 0 ? *reinterpret_cast<const f4*>(coly - p.nx) : cen;
    const f4 dn = oy < p.gny - 1 ? *reinterpret_cast<const f4*>(coly + p.nx) : cen;
    f4 odx, ody, osl, oas;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int x = ox + t;
        const float xm = t == 0 ? left : mid[t == 0 ? 0 : t - 1], xp = t == 3 ? right : mid[t == 3 ? 3 : t + 1];
        // numpy.gradient: central difference / 2 inside, first-order one-sided at the edges
        float dx;
        if (x == 0) dx = mid[1] - mid[0];
        else if (x == p.nx - 1) dx = mid[3] - mid[2];
        else dx = (xp - xm) * 0.5f;
        float dy;
        if (oy == 0) dy = dn[t] - cen[t];
        else if (oy == p.gny - 1) dy = cen[t] - up[t];
        else dy = (dn[t] - up[t]) * 0.5f;
        float rx, ry, slope, aspect;
        resolution_at(p, oy, x, rx, ry);
        gradient_values(dx, dy, rx, ry, slope, aspect);
        odx[t] = dx;
        ody[t] = dy;
        osl[t] = slope;
        oas[t] = aspect;
    }
    const size_t o = (size_t)(oy - p.out_row0) * p.nx + ox;
    // streaming stores (nobody reads the four planes back from the caches; the smoothed rows above and below are what
    // L2 should keep): gradient sigma 3.25 / 30.25 on 32768^2 5.95-6.38 / 9.12-9.30 -> 5.87-5.90 / 8.96-9.00 ms
    if (p.dx) __builtin_nontemporal_store(odx, reinterpret_cast<f4*>(p.dx + o));
    if (p.dy) __builtin_nontemporal_store(ody, reinterpret_cast<f4*>(p.dy + o));
    if (p.slope) __builtin_nontemporal_store(osl, reinterpret_cast<f4*>(p.slope + o));
    if (p.aspect) __builtin_nontemporal_store(oas, reinterpret_cast<f4*>(p.aspect + o));
}

__device__ __forceinline__ double raw_reflect(const GradArgs& p, int gy, int gx) {
    gy = reflect_index(gy, p.gny);
    gx = reflect_index(gx, p.nx);
    return (double)p.raw[(size_t)(gy - p.raw_row0) * p.nx + gx];
}

// Sobel pair / 8, true convolution (kernel flipped), evaluated in float64 like ndimage.
template <bool FINISH>
__global__ __launch_bounds__(kThreads) void sobel_kernel(GradArgs p) {
    const int ox = blockIdx.x * kThreads + threadIdx.x;
    const int oy = p.out_row0 + blockIdx.y;
    if (ox >= p.nx) return;
    const double nw = raw_reflect(p, oy - 1, ox - 1), n = raw_reflect(p, oy - 1, ox),
                 ne = raw_reflect(p, oy - 1, ox + 1);
    const double w = raw_reflect(p, oy, ox - 1), e = raw_reflect(p, oy, ox + 1);
    const double sw = raw_reflect(p, oy + 1, ox - 1), s = raw_reflect(p, oy + 1, ox),
                 se = raw_reflect(p, oy + 1, ox + 1);
    // float32 taps 1/8, 2/8 are exact; accumulate in double, round once
    const float dx = (float)(0.125 * (ne - nw) + 0.25 * (e - w) + 0.125 * (se - sw));
    const float dy = (float)(0.125 * (sw - nw) + 0.25 * (s - n) + 0.125 * (se - ne));
    if (FINISH) {
        finish_gradient(p, oy, ox, dx, dy);
    } else {
        const size_t o = (size_t)(oy - p.out_row0) * p.nx + ox;
        if (p.dx) p.dx[o] = dx;
        if (p.dy) p.dy[o] = dy;
    }
}

// ---- axis 1 fused with the gradient epilogue -------------------------------------------------
// Same tiling as gauss_axis1_kernel, but the 64 x (NW*TB) smoothed tile stays in LDS and the
// block emits dx, dy, slope, aspect for the 62 x (NW*TB - 2) pixels inside it: the smoothed
// plane never goes to HBM (28 instead of 36 B/pixel for the whole gradient).  Tiles sit on
// global multiples of 62 rows / (NW*TB - 2) columns, so results do not depend on the row block.
//
// Blocks are persistent and walk the tile list in row-major order.  A stamped build of the
// one-tile-per-block version showed a wave spending 42 % of its time fetching its tile and 14 % in
// the convolution, so the next tile's samples are loaded into registers as soon as the current
// tile is in LDS and stay in flight through the convolution and the epilogue.  The barriers are raw
// s_barrier + lgkmcnt(0): __syncthreads() would also wait for those loads.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int TB, int KB, int NW, int PF>
__global__ __launch_bounds__(NW * 64) void gauss_axis1_grad_kernel(GaussArgs p, GradArgs g, int tiles_x, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    constexpr int TR = 64;
    constexpr int TC = NW * TB;
    constexpr int OUT_R = TR - 2, OUT_C = TC - 2;
    constexpr int RPW = TR / NW;  // tile rows fetched per wave
    // PF: samples per lane and row held in registers for the next tile, cols_l <= 64 PF (the launcher picks it)
    const int R = p.radius;
    const int cols_l = TC + p.nchunks * KB - 1;
    const int stride = cols_l | 1;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    float pf[RPW][PF];
    auto issue = [&](int tile) {
        const int sx0 = (tile % tiles_x) * OUT_C - 1;                                   // smoothed tile origin
        const int sy0 = (g.out_row0 / OUT_R + tile / tiles_x) * OUT_R - 1;
        const bool inner = sx0 - R >= 0 && sx0 - R + cols_l <= p.nx;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave + rr * NW;
            const int gy = min(max(sy0 + r, p.in_row0), p.in_row0 + p.in_rows - 1);  // unused rows clamp
            const float* src = p.in + (size_t)(gy - p.in_row0) * p.nx;
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int k = min(lane + 64 * q, cols_l - 1);
                pf[rr][q] = inner ? src[sx0 - R + k] : src[reflect_index(sx0 - R + k, p.nx)];
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            float* dst = L + (wave + rr * NW) * stride;
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int k = lane + 64 * q;
                if (k < cols_l) dst[k] = pf[rr][q];
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int ox0 = (tile % tiles_x) * OUT_C;                                   // first output column
        const int oy0 = (g.out_row0 / OUT_R + tile / tiles_x) * OUT_R;               // first output row (global)
        lds_barrier();  // the previous tile's epilogue is done with the LDS image
        commit();
        lds_barrier();
        if (tile + (int)gridDim.x < ntiles) issue(tile + (int)gridDim.x);

        const float* rowp = L + lane * stride + wave * TB;
        auto fetch = [&](int i) -> float { return rowp[i]; };
        const float c = finite_or_zero(rowp[R]);
        float acc[TB];
        tap_chunks<TB, KB>(p.taps, p.nchunks, c, fetch, acc);
        lds_barrier();
        constexpr int ostride = TC + 1;
        float* O = L;  // smoothed tile, rounded to float32 like the reference's intermediate
#pragma unroll
        for (int t = 0; t < TB; ++t) O[lane * ostride + wave * TB + t] = c + acc[t];
        lds_barrier();

        for (int idx = threadIdx.x; idx < OUT_R * OUT_C; idx += NW * 64) {
            const int r = idx / OUT_C, k = idx % OUT_C;
            const int oy = oy0 + r, ox = ox0 + k;
            if (oy < g.out_row0 || oy >= g.out_row0 + g.out_rows || ox >= g.nx) continue;
            const float* q = O + (r + 1) * ostride + (k + 1);  // smoothed value of this pixel
            // numpy.gradient: central difference / 2 inside, first-order one-sided at the edges
            float dx, dy;
            if (ox == 0) dx = q[1] - q[0];
            else if (ox == g.nx - 1) dx = q[0] - q[-1];
            else dx = (q[1] - q[-1]) * 0.5f;
            if (oy == 0) dy = q[ostride] - q[0];
            else if (oy == g.gny - 1) dy = q[0] - q[-ostride];
            else dy = (q[ostride] - q[-ostride]) * 0.5f;
            finish_gradient(g, oy, ox, dx, dy);
        }
    }
}

// ---- axis 1 by wavefront shifts, fused with the gradient epilogue: no LDS at all -----------------
// Lanes own GC = 16 adjacent columns of one row (a wave-row is 1024 columns, 4 KiB, loaded with
// four coalesced dwordx4 per lane).  out[16L+t] = sum_k w[k] in[16L+t+k-R] is evaluated like the disc
// chain of disc_wave_impl.hpp: for each lane offset D the lane adds the contribution of its own 16
// samples to 16 partial sums (256 FMAs against 31 wave-uniform taps), and the partial sums hop
// one lane per step with a DPP wave shift.  The three most recent smoothed rows stay in
// registers, so the central differences, slope and aspect of a row are formed without the smoothed
// plane ever leaving the register file, and the four outputs leave as 64-byte pieces per lane.
//
// float32 accumulation keeps its digits through offsets: samples enter relative to the lane's own
// first sample, and that sample's offset from a wave-uniform base rides along as a 17th input with
// the summed taps (wsum), so every product is a small number.
constexpr int GC = 16;

struct WaveGradArgs {
    const float* in;   // plane already smoothed along axis 0: rows [in_row0, in_row0 + in_rows)
    int in_row0, in_rows;
    const float* wtab;  // nsteps x 32: taps w[GC*D + (j - 15) + R] for j = 0..30, zero outside the filter
    const float* wsum;  // nsteps x 16: for output sub-column t, sum over s of the taps applied to sample s
    int radius, d_lo, nsteps, nvl;
    int out_c;          // output columns per strip = GC * (nvl - 2)
    int rows_per_wave;
    float* smooth_out;  // non-NULL: store the smoothed rows themselves instead of the gradient
    GradArgs g;
};

#define GDPP_SHL1 0x130  // lane i takes lane i + 1
#define GDPP_SHR1 0x138  // lane i takes lane i - 1
__device__ __forceinline__ float hop_down(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), GDPP_SHL1, 0xf, 0xf, true));
}
__device__ __forceinline__ float hop_up(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), GDPP_SHR1, 0xf, 0xf, true));
}

__device__ __forceinline__ void wave_row_load(const WaveGradArgs& p, int gy, int xs, bool inner, float (&v)[GC]) {
    const int by = min(max(gy - p.in_row0, 0), p.in_rows - 1);  // rows outside only feed unused outputs
    const float* row = p.in + (size_t)by * p.g.nx;
    if (inner) {
#pragma unroll
        for (int q = 0; q < GC / 4; ++q) {
            const float4 f = *reinterpret_cast<const float4*>(row + xs + 4 * q);
            v[4 * q] = f.x;
            v[4 * q + 1] = f.y;
            v[4 * q + 2] = f.z;
            v[4 * q + 3] = f.w;
        }
    } else {
#pragma unroll
        for (int s = 0; s < GC; ++s) v[s] = row[reflect_index(xs + s, p.g.nx)];
    }
}

// smoothed values of one row: lane l ends up with the 16 columns of lane l - d_lo (valid l < nvl).
// Inputs enter the chain as small numbers only: x[s] = v[s] - v[0] (relief inside the lane's 16
// columns) and eps = v[0] - v[0] of the lane to the left, weighted with the cumulated taps wcum, which
// is an exact rewrite of sum_D wsum[D] (v0[L+D] - v0[L]); v0[L] itself is added once at the end.
__device__ __forceinline__ void wave_row_smooth(const WaveGradArgs& p, int lane, const float (&v)[GC],
                                                float (&out)[GC]) {
    const float eps = v[0] - hop_up(v[0]);
    float x[GC];
#pragma unroll
    for (int s = 0; s < GC; ++s) x[s] = v[s] - v[0];
    float acc[GC];
#pragma unroll
    for (int t = 0; t < GC; ++t) acc[t] = 0.0f;
    for (int step = 0; step < p.nsteps; ++step) {
        // Wave-uniform tap tables, read through the constant address space: the kernel stores to
        // global memory between uses, so as plain global pointers the compiler does not treat these
        // loads as invariant and fetched all 47 taps of a step through the vector path into VGPRs
        // (3.9e8 vector loads per 32768^2 launch, in a kernel at 246 VGPRs).  Nothing writes the
        // tables while the kernel runs.
        typedef const __attribute__((address_space(4))) float* cptr;
        cptr w = (cptr)p.wtab + step * 32;
        cptr wc = (cptr)p.wsum + step * GC;
#pragma unroll
        for (int t = 0; t < GC; ++t) {
            float a = step == 0 ? 0.0f : hop_down(acc[t]);
            a = fmaf(wc[t], eps, a);
#pragma unroll
            for (int sidx = 0; sidx < GC; ++sidx) a = fmaf(w[sidx - t + 15], x[sidx], a);
            acc[t] = a;
        }
    }
    // first sample of the lane whose columns these sums belong to
    const int src = min(lane - p.d_lo, 63);
    const float base = __int_as_float(__builtin_amdgcn_ds_bpermute(src * 4, __float_as_int(v[0])));
#pragma unroll
    for (int t = 0; t < GC; ++t) out[t] = acc[t] + base;
}

__global__ __launch_bounds__(kThreads) void gauss_axis1_wave_grad_kernel(WaveGradArgs p) {
    const GradArgs& g = p.g;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ox0 = blockIdx.x * p.out_c;                      // first output column of the strip
    const int xs = ox0 - GC * (1 - p.d_lo) + GC * lane;        // first staged column of this lane
    const int col0 = ox0 + GC * (lane - 1);                    // first column this lane's results cover
    const bool inner = ox0 - GC * (1 - p.d_lo) >= 0 && ox0 - GC * (1 - p.d_lo) + 64 * GC <= g.nx &&
                       (g.nx & 3) == 0;
    const int tile = (g.out_row0 / p.rows_per_wave) + blockIdx.y * (kThreads / 64) + wave;
    const int y_begin = max(tile * p.rows_per_wave, g.out_row0);
    const int y_end = min((tile + 1) * p.rows_per_wave, g.out_row0 + g.out_rows);
    if (y_begin >= y_end) return;
    const bool lane_out = lane >= 1 && lane <= p.nvl - 2 && col0 < g.nx;
    const bool vec_ok = (g.nx & 3) == 0 && col0 + GC <= g.nx;
    float* planes[4] = {g.dx, g.dy, g.slope, g.aspect};

    float raw[GC], up[GC], mid[GC], dn[GC];
#pragma unroll
    for (int t = 0; t < GC; ++t) up[t] = mid[t] = 0.0f;
    wave_row_load(p, y_begin - 1, xs, inner, raw);
    // rows y_begin-1 .. y_end are smoothed one after the other; once three of them are there the
    // middle one is differenced and written
#pragma unroll 1
    for (int yy = y_begin - 1; yy <= y_end; ++yy) {
        wave_row_smooth(p, lane, raw, dn);
        if (yy < y_end) wave_row_load(p, yy + 1, xs, inner, raw);  // in flight during the epilogue
        const int y = yy - 1;  // row held in `mid`
        if (y >= y_begin) {
            // neighbours across the lane boundary
            const float left = hop_up(mid[GC - 1]);
            const float right = hop_down(mid[0]);
            if (lane_out && p.smooth_out) {
                const size_t o = (size_t)(y - g.out_row0) * g.nx + col0;
#pragma unroll
                for (int t = 0; t < GC; ++t)
                    if (col0 + t < g.nx) p.smooth_out[o + t] = mid[t];
            } else if (lane_out) {
                const size_t o = (size_t)(y - g.out_row0) * g.nx + col0;
#pragma unroll
                for (int q = 0; q < GC / 4; ++q) {
                    float val[4][4];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const int t = 4 * q + tt;
                        const int ox = col0 + t;
                        const float qm = t == 0 ? left : mid[t == 0 ? 0 : t - 1];
                        const float qp = t == GC - 1 ? right : mid[t == GC - 1 ? GC - 1 : t + 1];
                        // numpy.gradient: central difference / 2 inside, one-sided at the edges
                        float dx, dy;
                        if (ox == 0) dx = qp - mid[t];
                        else if (ox == g.nx - 1) dx = mid[t] - qm;
                        else dx = (qp - qm) * 0.5f;
                        if (y == 0) dy = dn[t] - mid[t];
                        else if (y == g.gny - 1) dy = mid[t] - up[t];
                        else dy = (dn[t] - up[t]) * 0.5f;
                        float rx, ry;
                        resolution_at(g, y, min(ox, g.nx - 1), rx, ry);
                        gradient_values(dx, dy, rx, ry, val[2][tt], val[3][tt]);
                        val[0][tt] = dx;
                        val[1][tt] = dy;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (!planes[k]) continue;
                        if (vec_ok) {
                            *reinterpret_cast<float4*>(planes[k] + o + 4 * q) =
                                make_float4(val[k][0], val[k][1], val[k][2], val[k][3]);
                        } else {
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt)
                                if (col0 + 4 * q + tt < g.nx) planes[k][o + 4 * q + tt] = val[k][tt];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < GC; ++t) {
            up[t] = mid[t];
            mid[t] = dn[t];
        }
    }
}

// ---- plane transpose (used to run very long axis-1 filters as axis-0 filters) -----------------
__global__ __launch_bounds__(kThreads) void transpose_kernel(const float* in, float* out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int y = y0 + ty + k, x = x0 + tx;
        if (y < rows && x < cols) tile[ty + k][tx] = in[(size_t)y * cols + x];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int y = x0 + ty + k, x = y0 + tx;  // coordinates in the transposed plane
        if (y < cols && x < rows) out[(size_t)y * rows + x] = tile[tx][ty + k];
    }
}

// ---- long filters on the matrix cores: banded Toeplitz x data with v_mfma_f32_32x32x2_f32 --------------
// out = T in, T[i][k] = w[k - i] for 0 <= k - i <= 2 R.  For a tile of 32 outputs along the filter axis the
// band touches K = 32 + 2 R inputs, so the product spends (32 + 2 R) / (2 R + 1) of the minimum number of
// multiply-adds (1.13 at radius 121) - but at the f32 MFMA rate, which one wave per SIMD reaches with
// nothing but an LDS read per operand, where the vector-ALU kernels issue 2.3 instructions per useful FMA.
// The result of a 32x32x2 f32 MFMA chain is the k-ordered fmaf chain, one rounding per product; the taps
// outside the band are exact zeros, so an output is the fmaf chain over its own 2 R + 1 taps in ascending
// order whatever tile it falls in (partition invariance needs no tile alignment; only the accumulation
// offsets do: they are taken per column / per row from samples that any block covering the output holds).
// A non-finite sample reaches every output of the tiles whose (padded) band contains it (0 x NaN), i.e. up to 37
// rows / columns further than in ndimage.gaussian_filter.  Two accumulators (even / odd groups) were tried to
// take the dependent-MFMA wait out: 7.9 ms against 6.9 ms for the one chain, so it stays one chain.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// The 8 LDS reads of the next phase go out two behind each of the first 4 MFMAs of this one (with the vector-ALU
// instruction that forms their address and the scalar ring arithmetic), so the last of them has 4 MFMAs (260
// cycles) to land before the next phase subtracts the offsets from it; then a fence.
__device__ __forceinline__ void spread_behind_mfmas() {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void mfma_operands_ready() {
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
}
// a - c on a register pair in one instruction (the compiler splits the vector subtraction into two v_add_f32
// whenever it likes the register allocation better that way)
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 c) {
    // The hazard recogniser does not look inside inline assembly, and an MFMA that reads the pair one wait state
    // after this instruction gets stale data (seen: wrong results with `v_pk_add_f32; s_nop 0; v_mfma`).  The
    // callers issue the four subtractions of a phase, then mfma_operands_ready(), then the MFMAs.
    f32x2 o;
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(a), "v"(c));
    return o;
}
constexpr int kMfmaCols = 128;  // axis 0: columns per block (4 waves x 32)

// wz[n] = w[n - 31] (0 outside the filter), n in [0, K + 32): lane (i = l & 31, h = l >> 5) of step s reads
// wz[2 s + h - i + 31], the Toeplitz entry T[i][2 s + h]
constexpr int kWzPad = 80;  // zeros behind the taps: what the padded steps and the fetches two groups ahead read past the band (<= 60)
__device__ __forceinline__ void fill_toeplitz_table(float* wz, const float* taps, int R, int K) {
    for (int n = threadIdx.x; n < K + kWzPad; n += blockDim.x) wz[n] = (n >= 31 && n - 31 <= 2 * R) ? taps[n - 31] : 0.0f;
}

// (the float32 matrix-core kernels of round 2 - v_mfma_f32_32x32x2_f32, 11.3 ms at sigma 30.25 - were retired in round 5:
// docs/DESIGN_HISTORY.md, profiles/r02_gauss_mfma.txt, r03_gauss_f16.txt keep their numbers)

// ---- the banded products on the f16 matrix pipe (round 3) ------------------------------------------------
// v_mfma_f32_32x32x16_f16 does 16 x the multiply-adds of the f32 instruction per cycle.  A float32 difference
// d = (x - c) / 4 (c: the tile's offset as above; the quarter keeps |d| inside f16 for any |x| <= 1e5) is split
// on the fly into d = h + l + e, h = d cut to 11 significant bits (exact in f16), l = f16(d - h) rounded to
// nearest, |e| <= 2^-22 |d|; a tap t 2^k (k: the scale that puts the largest tap in [512, 1024)) into th + tm the
// same way, once per kernel, in registers.  Three products stand for t d: th h + tm h + th l (the fourth, tm l, is
// <= 2^-22 t d like e).  The sums are float32 inside the matrix pipe, 3 roundings per 16 taps instead of 16, on
// values of the size of x - c: the error against the exact filter is that of the f32 chain or smaller
// (tools/gauss_f16_error.py).  x = c over a lake gives h = l = 0 and the output c, as before.
// What changes is the order of the additions: 16 taps at a time, on a grid of steps that starts Rp = R rounded up
// to 16 above (left of) the 32-output tile.  Tiles sit on global multiples of 32 on both axes, so an output still
// sees the same operands in the same places whatever row block it is computed in.
// Non-finite and absurd samples (|x| > 1e5) never reach the matrix pipe: the loader stages them as 0, remembers
// that the window holds one, and the tile is marked; the repair pass then recomputes, tap by tap in float32, the
// outputs whose OWN window holds such a sample and leaves the others alone (they are what they would be without
// it: a zero tap times 0 or times a finite sample adds the same nothing).  The mark may be conservative (a block's
// clamped ghost rows, neighbouring columns): it costs time, never bits.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr float kWild = 1.0e5f;
__device__ __forceinline__ bool wild(float x) { return !(fabsf(x) <= kWild); }

__device__ __forceinline__ f32x2 pk_fma_pure(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 o;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}
__device__ __forceinline__ f32x2 pk_sub_pure(f32x2 a, f32x2 c) {
    f32x2 o;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(a), "v"(c));
    return o;
}

// lane (j = lane & 31, g = lane >> 5) of step s holds the taps T[j][16 s + 8 g + q], q = 0 ... 7: the operand the
// Toeplitz factor has on either side of the product (A for axis 0, B for axis 1: the band is symmetric)
template <int S>
__device__ __forceinline__ void build_tap_blocks(const float* taps, int R, float scale, int lane, f16x8 (&hi)[S], f16x8 (&lo)[S]) {
    const int j = lane & 31, g = lane >> 5, Rp = 8 * (S - 2);
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = 16 * s + 8 * g + q - j - (Rp - R);
            const float t = (n >= 0 && n <= 2 * R) ? taps[n] * scale : 0.0f;
            const _Float16 th = (_Float16)t;
            hi[s][q] = th;
            lo[s][q] = (_Float16)(t - (float)th);
        }
    }
}

// 8 samples -> (x - c) / 4 -> f16 pair of operands
__device__ __forceinline__ void split8(const float (&x)[8], f32x2 quarter, f32x2 mcq, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        // (plain fma / sub: a wave alone on its SIMD takes 53 ns for the 8 samples this way, 64 ns with v_pk_fma_f32 /
        // v_pk_add_f32: tools/ubench/split_rate.hip)
        const f32x2 d = {__builtin_fmaf(x[2 * u], quarter[0], mcq[0]), __builtin_fmaf(x[2 * u + 1], quarter[1], mcq[1])};
        // (as one vector operation: __builtin_bit_cast on the ELEMENT d[1] reads d[0] with this compiler)
        const f32x2 hf = __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2, d) & u32x2{0xFFFFE000u, 0xFFFFE000u});
        const f32x2 r = {d[0] - hf[0], d[1] - hf[1]};
        const f16x2 hh = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(hf[0], hf[1]));  // exact: hf has 11 bits
        const f16x2 ll = __builtin_convertvector(r, f16x2);                                     // v_cvt_pk_f16_f32, to nearest
        hi[2 * u] = hh[0];
        hi[2 * u + 1] = hh[1];
        lo[2 * u] = ll[0];
        lo[2 * u + 1] = ll[1];
    }
}

template <bool DATA_IS_A, int NP>
__device__ __forceinline__ void f16_products(const f16x8& th, const f16x8& tm, const f16x8& dh, const f16x8& dl, f32x16& acc) {
    if (DATA_IS_A) {
        if (NP >= 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(dl, tm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh, tm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(dl, th, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh, th, acc, 0, 0, 0);
    } else {
        if (NP >= 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tm, dl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tm, dh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, dl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, dh, acc, 0, 0, 0);
    }
}

constexpr int kNoWild = -(1 << 28);

// Axis 0.  A wave owns 32 columns and a run of row tiles, top to bottom, and talks to nobody: its window of input
// rows is a ring in LDS (its 32 columns of the block's [row][128] array), the rows the next tile adds are loaded
// into registers while this tile is computed and written over the oldest rows afterwards (the wave's own reads come
// first in program order: no spare rows, no barrier).
// MT = 2: the wave's tile is 64 rows (two MFMA tiles on top of each other, global multiples of 64) with ONE offset
// per column, so one split of the samples feeds both: the vector-ALU work per output halves.  The offset row is 32
// rows into the tile; a row block holds it for every tile it computes when it carries 32 ghost rows, so the
// launcher takes MT = 2 from radius 32 (except behind a fused launch, whose bands are 32-row tiles).
template <int S, int NP, int MT>
__global__ __launch_bounds__(256) void gauss_axis0_f16_kernel(GaussArgs p, int tile_first, int ntiles, int tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    constexpr int Rp = 8 * (S - 2), NS = S + 2 * (MT - 1), RR = 16 * NS, TILE = 32 * MT;
    constexpr int kStay = (RR + TILE - 1) / TILE;  // tiles whose window holds a slab of TILE rows, rounded up
    static_assert(S % 2 == 0, "whole 16-row steps on both sides of the tile");
    const int R = p.radius;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, g = lane >> 5;
    const int x0 = blockIdx.x * kMfmaCols;
    const int tb = blockIdx.y * tiles_per_block, te = min(tb + tiles_per_block, ntiles);
    if (tb >= te) return;
    if (p.run_if && *p.run_if == 0) return;
    f16x8 twh[S], twl[S];
    build_tap_blocks<S>(p.taps, R, p.tap_scale, lane, twh, twl);
    float* ring = L + 32 * wave;  // [row][128], this wave's 32 columns
    // loader: lane -> row (lane >> 3) of a pass of 8 rows, 4 columns (lane & 7) * 4
    const int lq = lane >> 3, lcol = (lane & 7) * 4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    // the lane's 4 columns of a row: one 16-byte load (dword alignment is all it needs, so any row length will do); the
    // group that straddles the DEM's right edge (widths that are not multiples of 4) and the groups beyond it go
    // column by column, clamped (what lies past the edge is never stored)
    const int lx = x0 + 32 * wave + lcol;
    const bool lane_full = lx + 4 <= p.nx;
    auto load4 = [&](const float* rowbase) {
        if (lane_full) return *reinterpret_cast<const f4*>(rowbase + lx);
        f4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rowbase[min(lx + e, p.nx - 1)];
        return v;
    };
    auto row_base = [&](int gy) {
        gy = reflect_index(gy, p.gny);
        gy = min(max(gy, p.in_row0), p.in_row0 + p.in_rows - 1);
        return p.in + (size_t)(gy - p.in_row0) * p.nx;
    };
    float* const ring_lane = ring + lq * kMfmaCols + lcol;
    const int row_lo = max(0, p.in_row0), row_hi = min(p.gny, p.in_row0 + p.in_rows);
    int last_wild = kNoWild;  // last tile whose window holds a staged sample that is not a plain finite one
    {
        const int y0 = (tile_first + tb) * TILE;
        bool bad = false;
        for (int k = 0; k < RR; k += 8) {
            f4 v = load4(row_base(y0 - Rp + k + lq));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool w = wild(v[e]);
                bad |= w;
                v[e] = w ? 0.0f : v[e];
            }
            *reinterpret_cast<f4*>(ring_lane + k * kMfmaCols) = v;
        }
        if (__builtin_amdgcn_ballot_w64(bad)) last_wild = tb + kStay - 1;
    }
    int base = 0;  // ring row of input row y0 - Rp
    const int xw = 32 * wave + j;
    const unsigned out_lane_off = (unsigned)(4 * g * p.nx + xw) * 4u;
    const f32x2 quarter = {0.25f, 0.25f};
    const float* const bl = L + (8 * g) * kMfmaCols + xw;
    for (int t = tb; t < te; ++t) {
        const int y0 = (tile_first + t) * TILE;
        f4 pre[4 * MT];
        const bool more = t + 1 < te;
        if (more) {
            const int n0 = y0 - Rp + RR;  // the rows the next tile adds
            if (n0 >= row_lo && n0 + TILE <= row_hi) {
                const float* rb = p.in + (size_t)(n0 + lq - p.in_row0) * p.nx;
#pragma unroll
                for (int q = 0; q < 4 * MT; ++q) pre[q] = load4(rb + (size_t)(8 * q) * p.nx);
            } else {
#pragma unroll
                for (int q = 0; q < 4 * MT; ++q) pre[q] = load4(row_base(n0 + 8 * q + lq));
            }
        }
        int sc = base + Rp + TILE / 2;
        sc = sc >= RR ? sc - RR : sc;
        const float c = L[sc * kMfmaCols + xw];  // the column's sample at the tile's middle row (0 if it was not a plain finite one)
        const f32x2 mcq = {-0.25f * c, -0.25f * c};
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[m][v] = 0.0f;
        int slot = base;
        auto fetch = [&](float (&x)[8]) {
            const float* bp = bl + slot * kMfmaCols;
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = bp[q * kMfmaCols];
            slot += 16;
            slot = slot >= RR ? slot - RR : slot;
        };
        float xa[8], xb[8];
        fetch(xa);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            f16x8 dh, dl;
            // the next step's samples are requested before this step's are split (the compiler otherwise sinks the
            // reads to their use and the wave, alone on its SIMD, waits out the LDS latency every step)
            if (s & 1) {
                if (s + 1 < NS) fetch(xa);
                __builtin_amdgcn_sched_barrier(0);
                split8(xb, quarter, mcq, dh, dl);
            } else {
                if (s + 1 < NS) fetch(xb);
                __builtin_amdgcn_sched_barrier(0);
                split8(xa, quarter, mcq, dh, dl);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (s - 2 * m >= 0 && s - 2 * m < S) f16_products<false, NP>(twh[s - 2 * m], twl[s - 2 * m], dh, dl, acc[m]);
        }
        if (more) {
            bool bad = false;
#pragma unroll
            for (int q = 0; q < 4 * MT; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) bad |= wild(pre[q][e]);
            if (__builtin_amdgcn_ballot_w64(bad)) {
#pragma unroll
                for (int q = 0; q < 4 * MT; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) pre[q][e] = wild(pre[q][e]) ? 0.0f : pre[q][e];
                last_wild = t + kStay;  // the rows enter with tile t + 1
            }
            int s8 = base;  // over the oldest rows
#pragma unroll
            for (int q = 0; q < 4 * MT; ++q) {
                *reinterpret_cast<f4*>(ring_lane + s8 * kMfmaCols) = pre[q];
                s8 += 8;
                s8 = s8 >= RR ? s8 - RR : s8;
            }
        }
        const int ox = x0 + xw;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int ym = y0 + 32 * m;
            if (lane == 0) p.flags[((size_t)(t * MT + m) * gridDim.x + blockIdx.x) * 4 + wave] = last_wild >= t ? 1 : 0;
            if (ox < p.nx) {
                if (ym >= p.out_row0 && ym + 32 <= p.out_row0 + p.out_rows) {
                    char* ub = reinterpret_cast<char*>(p.out + (size_t)(ym - p.out_row0) * p.nx + x0);
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        *reinterpret_cast<float*>(ub + (size_t)((v & 3) + 8 * (v >> 2)) * p.nx * 4 + out_lane_off) = fmaf(acc[m][v], p.out_scale, c);
                } else {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int oy = ym + (v & 3) + 8 * (v >> 2) + 4 * g;
                        if (oy >= p.out_row0 && oy < p.out_row0 + p.out_rows) p.out[(size_t)(oy - p.out_row0) * p.nx + ox] = fmaf(acc[m][v], p.out_scale, c);
                    }
                }
            }
        }
        base += TILE;
        base = base >= RR ? base - RR : base;
    }
}

// Axis 1.  A wave owns a band of 32 rows and marches along x; its window of input columns is a ring of LDS columns
// (row pitch = window + 4 floats: the two 16-byte reads of an A operand are aligned and 16 rows fall into 16
// different bank groups).  `in` holds plane rows [0, rows); a row's result depends on that row alone.  MT = 2: tiles
// of 64 columns with one offset per row (the sample 32 columns into the tile: always there), one split for two MFMA
// tiles.  NW waves per block: the widest window (radius 113 ... 121, MT = 2) fits LDS three times, not four.
template <int S, int NP, int MT, int NW>
__global__ __launch_bounds__(64 * NW) void gauss_axis1_f16_kernel(GaussArgs p, int rows, int nseg) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    constexpr int Rp = 8 * (S - 2), NS = S + 2 * (MT - 1), RC = 16 * NS, pitch = RC + 4, TILE = 32 * MT;
    constexpr int kStay = (RC + TILE - 1) / TILE;
    static_assert(S % 2 == 0, "whole 16-column steps on both sides of the tile");
    const int R = p.radius;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* ring = L + wave * (32 * pitch + 32);
    float* crow = ring + 32 * pitch;
    const int gw = blockIdx.x * NW + wave;
    const int band = gw / nseg, seg = gw - band * nseg;
    const int r0 = band * 32;
    if (r0 >= rows) return;
    if (p.run_if && *p.run_if == 0) return;
    const int i = lane & 31, g = lane >> 5;
    f16x8 twh[S], twl[S];
    build_tap_blocks<S>(p.taps, R, p.tap_scale, lane, twh, twl);
    auto load_cols = [&](int xfirst, int q) {  // rows 2 q + g, columns xfirst + i
        const int r = min(r0 + 2 * q + g, rows - 1);
        const int cx = reflect_index(xfirst + i, p.nx);
        return p.in[(size_t)r * p.nx + cx];
    };
    const unsigned in_lane_off = (unsigned)(g * p.nx + i) * 4u;
    const unsigned out_lane_off = (unsigned)(4 * g * p.nx + i) * 4u;
    const bool full_band = r0 + 32 <= rows;
    const int ntile = (p.nx + TILE - 1) / TILE, nunit = (p.nx + 31) / 32;
    const int tper = (ntile + nseg - 1) / nseg;
    const int seg_first = seg * tper, seg_last = min(seg_first + tper, ntile);
    if (seg_first >= seg_last) return;
    // (the bands do not walk the row in step: gauss_axis1_s1_kernel has the reasons)
    const int turn = p.group0 != 0 && ntile >= 32 ? (int)(((unsigned)band * 40503u) % (unsigned)ntile) : 0;
    const int run_first = (seg_first + turn) % ntile, run_len = seg_last - seg_first;
    for (int part = 0; part < 2; ++part) {
    const int t_first = part == 0 ? run_first : 0;
    const int t_last = part == 0 ? min(run_first + run_len, ntile) : run_first + run_len - ntile;
    if (t_first >= t_last) continue;
    int last_wild = kNoWild;
    {
        bool bad = false;
        for (int k0 = 0; k0 < RC; k0 += 32) {
            if (k0 + i < RC) {
#pragma unroll 4
                for (int q = 0; q < 16; ++q) {
                    const float v = load_cols(t_first * TILE - Rp + k0, q);
                    const bool w = wild(v);
                    bad |= w;
                    ring[(2 * q + g) * pitch + k0 + i] = w ? 0.0f : v;
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(bad)) last_wild = t_first + kStay - 1;
    }
    int base = 0;  // ring column of input column x0 - Rp
    const f32x2 quarter = {0.25f, 0.25f};
    typedef float f4 __attribute__((ext_vector_type(4)));
    const float* const al = ring + i * pitch + 8 * g;
    for (int t = t_first; t < t_last; ++t) {
        const int x0 = t * TILE;
        float pre[16 * MT];
        const bool more = t + 1 < t_last;
        if (more) {
            const int n0 = x0 - Rp + RC;  // the columns the next tile adds
            if (full_band && n0 >= 0 && n0 + TILE <= p.nx) {
                const char* rb = reinterpret_cast<const char*>(p.in + (size_t)r0 * p.nx + n0);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int q = 0; q < 16; ++q) pre[16 * m + q] = *reinterpret_cast<const float*>(rb + (size_t)(2 * q) * p.nx * 4 + 128 * m + in_lane_off);
            } else {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int q = 0; q < 16; ++q) pre[16 * m + q] = load_cols(n0 + 32 * m, q);
            }
        }
        int sc = base + Rp + TILE / 2;
        sc = sc >= RC ? sc - RC : sc;
        const float c = ring[i * pitch + sc];  // the row's sample at the tile's middle column
        if (g == 0) crow[i] = c;
        const f32x2 mcq = {-0.25f * c, -0.25f * c};
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[m][v] = 0.0f;
        int slot = base;
        auto fetch = [&](float (&x)[8]) {
            const f4 lo4 = *reinterpret_cast<const f4*>(al + slot), hi4 = *reinterpret_cast<const f4*>(al + slot + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                x[q] = lo4[q];
                x[4 + q] = hi4[q];
            }
            slot += 16;
            slot = slot >= RC ? slot - RC : slot;
        };
        float xa[8], xb[8];
        fetch(xa);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            f16x8 dh, dl;
            if (s & 1) {
                if (s + 1 < NS) fetch(xa);
                __builtin_amdgcn_sched_barrier(0);
                split8(xb, quarter, mcq, dh, dl);
            } else {
                if (s + 1 < NS) fetch(xb);
                __builtin_amdgcn_sched_barrier(0);
                split8(xa, quarter, mcq, dh, dl);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (s - 2 * m >= 0 && s - 2 * m < S) f16_products<true, NP>(twh[s - 2 * m], twl[s - 2 * m], dh, dl, acc[m]);
        }
        if (more) {
            bool bad = false;
#pragma unroll
            for (int q = 0; q < 16 * MT; ++q) bad |= wild(pre[q]);
            if (__builtin_amdgcn_ballot_w64(bad)) {
#pragma unroll
                for (int q = 0; q < 16 * MT; ++q) pre[q] = wild(pre[q]) ? 0.0f : pre[q];
                last_wild = t + kStay;
            }
            int sl = base;  // over the oldest columns
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int q = 0; q < 16; ++q) ring[(2 * q + g) * pitch + sl + i] = pre[16 * m + q];
                sl += 32;
                sl = sl >= RC ? sl - RC : sl;
            }
        }
        float cr[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) cr[v] = crow[(v & 3) + 8 * (v >> 2) + 4 * g];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int xm = x0 + 32 * m, ox = xm + i;  // D: column = lane & 31
            if (xm >= p.nx) break;
            if (lane == 0) p.flags[(size_t)band * nunit + (t * MT + m)] = last_wild >= t ? 1 : 0;
            if (ox < p.nx) {
                if (full_band) {
                    char* ub = reinterpret_cast<char*>(p.out + (size_t)r0 * p.nx + xm);
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        *reinterpret_cast<float*>(ub + (size_t)((v & 3) + 8 * (v >> 2)) * p.nx * 4 + out_lane_off) = fmaf(acc[m][v], p.out_scale, cr[v]);
                } else {
                    float* o = p.out + (size_t)(r0 + 4 * g) * p.nx + ox;
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int ri = (v & 3) + 8 * (v >> 2);
                        if (r0 + 4 * g + ri < rows) o[(size_t)ri * p.nx] = fmaf(acc[m][v], p.out_scale, cr[v]);
                    }
                }
            }
        }
        base += TILE;
        base = base >= RC ? base - RC : base;
    }
    }
}


// ---- split once: axis 1, radius 49 ... 121 -------------------------------------------------------------------------
// The kernels above split every sample of a tile's window into its f16 pair again for every tile: 24 vector
// instructions per 8 samples and step (profiles/r03_gauss_f16.txt, section 5).  Here a sample is split ONCE, when it
// is staged: the ring in LDS holds the two f16 planes and a step is two 16-byte LDS reads and its MFMAs.  For that the
// offset a sample is taken against cannot belong to the tile: the row is cut into slabs of 64 columns (global
// multiples of 64) and every sample of a slab is taken against the slab's reference c_s, the sample of its row in
// the slab's first column.  A tile's window holds up to five slabs; with W_s[o] the sum of the taps that the output at
// position o of its own slab lays over slab s (a table of the host: float64 sums, rounded once) and b the tile's slab
//     out = c_b + [ scale * sum_k t_k (x_k - c_s(k)) / 4  +  sum_{s != b} (c_s - c_b) W_s ]
// - the taps sum to 1 as far as the tile kernels' "+ c" assumes it.  |x - c_s| is the relief inside 64 samples (the
// tile kernels: inside the 288 ... 320 of a window); the correction costs 2.5 vector instructions per output and slab.
// Results depend on the global slab grid only, not on how the march is cut.  Samples that are not plain finite ones go
// in as 0 and mark the tiles whose windows hold them (or hold samples taken against a reference that was one), as
// above: the repair pass recomputes those.  (Axis 0 was built the same way - a wave owns 16 columns, slab references
// on row 32 of each 64-row slab, R + 32 ghost rows for whole-DEM bits - and gives the tile kernel's time, 2.51 against
// 2.56 ms at sigma 30.25: 16 columns are 64-byte halves of lines shared with the neighbour wave, 1.56 x the plane
// fetched unless a block barrier per tile keeps the waves together, which takes the overlap away again; it is not in the
// library: profiles/r04_gauss_split_once.txt.  Round 6 built it again with staging waves loading whole 512-byte row
// segments for the block, and once more with every wave for itself at one wave per SIMD: correct, 5 - 8 % on axis 0 alone at
// radius 97 ... 121, nothing on the gradient, R + 32 ghost rows for every caller - left out again: profiles/r06_gauss_axis0_s1.txt,
// commit d700ebd.)
__device__ __forceinline__ int floor_div64(int a) { return a >> 6; }  // arithmetic shift: floor for negative a
// The ring's row pitch is the window + kS1Pad halves.  An A operand is one ds_read_b128 per plane: lane (n = lane & 15, kg =
// lane >> 4) reads 16 bytes of row n from position 8 kg.  The instruction is served in four groups of 16 lanes that are NOT
// the four kg - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) - i.e. rows
// {0-3, 12-15} at kg and rows {4-11} at kg + 1.  With the pitch of round 4 (window + 8 halves = 4 x 37 dwords at radius
// 121) row n starts 5 n sixteen-byte slots into the 256-byte bank row, and three of the eight rows of the second set land
// on slots of the first: every group two LDS cycles instead of one (SQ_LDS_BANK_CONFLICT 40 % of SQ_LDS_IDX_ACTIVE, rounds
// 4 and 5).  A pitch of 4 p dwords with p = 2 (mod 4) puts rows {0-3, 12-15} on the even slots and rows {4-11}, one slot
// further, on the odd ones: window + 16 halves for every window of 32 NK columns, NK odd.  (The time did not move - the
// conflicts were not what the kernel waits for: profiles/r06_gauss_axis0_s1.txt, section 4.)
constexpr int kS1Pad = 16;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// 4 samples -> (x - c) / 4 -> f16 pairs (split8's arithmetic)
__device__ __forceinline__ void split4(const float (&x)[4], float mcq, f16x4& hi, f16x4& lo) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const f32x2 d = {__builtin_fmaf(x[2 * u], 0.25f, mcq), __builtin_fmaf(x[2 * u + 1], 0.25f, mcq)};
        const f32x2 hf = __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2, d) & u32x2{0xFFFFE000u, 0xFFFFE000u});
        const f32x2 r = {d[0] - hf[0], d[1] - hf[1]};
        const f16x2 hh = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(hf[0], hf[1]));  // exact: hf has 11 bits
        const f16x2 ll = __builtin_convertvector(r, f16x2);                                     // to nearest
        hi[2 * u] = hh[0];
        hi[2 * u + 1] = hh[1];
        lo[2 * u] = ll[0];
        lo[2 * u + 1] = ll[1];
    }
}

// the Toeplitz factor of 16 outputs for v_mfma_f32_16x16x32_f16: lane (n = lane & 15, kg = lane >> 4) of step s holds
// the taps that window position j = 32 s + 8 kg + q, q = 0 ... 7, has for output n of sub-tile m (outputs 16 m + n of a
// 32-output tile whose window starts Rp before it): tap index j - n - 16 m - (Rp - R)
template <int NK>
__device__ __forceinline__ void build_tap_blocks16(const float* taps, int R, float scale, int lane, int m, f16x8 (&hi)[NK], f16x8 (&lo)[NK]) {
    const int n = lane & 15, kg = lane >> 4, Rp = 16 * (NK - 1);
#pragma unroll
    for (int s = 0; s < NK; ++s) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = 32 * s + 8 * kg + q - n - 16 * m - (Rp - R);
            const float t = (k >= 0 && k <= 2 * R) ? taps[k] * scale : 0.0f;
            const _Float16 th = (_Float16)t;
            hi[s][q] = th;
            lo[s][q] = (_Float16)(t - (float)th);
        }
    }
}

// data pair (dh, dl) x the tap pairs of two sub-tiles that share it (axis 1: the data is the A operand), the cross
// product with the taps' low part first (f16_products' order), the two chains alternating
template <int NP>
__device__ __forceinline__ void f16_products16_pair(const f16x8& t0h, const f16x8& t0l, const f16x8& t1h, const f16x8& t1l, const f16x8& dh,
                                                    const f16x8& dl, f32x4& acc0, f32x4& acc1) {
    if (NP >= 4) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(dl, t0l, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(dl, t1l, acc1, 0, 0, 0);
    }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(dh, t0l, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(dh, t1l, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(dl, t0h, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(dl, t1h, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(dh, t0h, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(dh, t1h, acc1, 0, 0, 0);
}

// a value of lane 8 r (the first of the 8 lanes that load row r) to all 8 of them
__device__ __forceinline__ float octet_first(float v, int lane) {
    const int q = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x00 /* quad_perm [0,0,0,0] */, 0xf, 0xf, false);
    const int up = __builtin_amdgcn_mov_dpp(q, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    return __builtin_bit_cast(float, (lane & 4) ? up : q);
}

// Axis 1.  A wave owns a band of 16 rows and marches along x in tiles of 32 columns - two sub-tiles of 16, each with
// its own tap blocks, both fed by one read of the window (v_mfma_f32_16x16x32_f16: the rate of the 32 x 32 x 16 form per
// multiply-add).  The ring - hi[16][PITCH], lo[16][PITCH] in f16, PITCH = window + 8 - is half the size of a 32-row
// band's, so EIGHT waves fit a CU, two per SIMD: one wave alone leaves half of the matrix pipe idle (a 32 x 32 x 16
// MFMA every 32 cycles from one wave, every 16 from two: tools/ubench/mfma_valu_overlap.hip) and hides at most 4
// vector instructions under each.  Window of the tile at x0: [x0 - Rp, x0 + 32 + Rp) = 32 NK columns, Rp = 16 (NK - 1)
// >= R.  The loader takes the 32 columns a tile adds two tiles ahead: lane -> rows (lane >> 3) and + 8, columns
// 4 (lane & 7) ...: 16-byte loads, 8 rows x 128 B per instruction.  The 32 columns lie in one slab; a slab starts
// with them when they start on a multiple of 64.  LDS: 8 x (ring + references of 8 slabs x 16 rows) + the table W.
// Flags: one byte per 16 x 32 tile, [band][tile]; the repair pass takes the two of a 32 x 32 unit together.
template <int NK, int NP>
__global__ __launch_bounds__(512) void gauss_axis1_s1_kernel(GaussArgs p, int rows, int nseg) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    constexpr int Rp = 16 * (NK - 1), RC = 32 * NK, PITCH = RC + kS1Pad, NSIDE = (Rp + 63) / 64;
    constexpr int kStay = NK;                         // tiles whose window holds a group of 32 columns
    constexpr int kWaveFloats = 16 * PITCH + 8 * 16;  // two f16 planes = 16 x PITCH floats, then the references
    static_assert(NSIDE <= 2 && Rp % 32 == 0, "at most 5 slabs per window; groups of 32 columns inside one slab");
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int R = p.radius;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* const wlds = L + 8 * kWaveFloats;  // W[5][64]
    for (int k = threadIdx.x; k < 5 * 64; k += 512) wlds[k] = p.wtab[k];
    __syncthreads();
    _Float16* const hi = reinterpret_cast<_Float16*>(L + wave * kWaveFloats);
    _Float16* const lo = hi + 16 * PITCH;
    float* const ctab = L + wave * kWaveFloats + 16 * PITCH;  // [8 slabs][16 rows]
    const int gw = blockIdx.x * 8 + wave;
    const int band = gw / nseg, seg = gw - band * nseg;
    const int r0 = band * 16;
    if (r0 >= rows) return;
    if (p.run_if && *p.run_if == 0) return;
    const int ntile = (p.nx + 31) / 32;
    const int tper = (ntile + nseg - 1) / nseg;
    const int t_first = seg * tper, t_last = min(t_first + tper, ntile);
    if (t_first >= t_last) return;
    f16x8 t0h[NK], t0l[NK], t1h[NK], t1l[NK];
    build_tap_blocks16<NK>(p.taps, R, p.tap_scale, lane, 0, t0h, t0l);
    build_tap_blocks16<NK>(p.taps, R, p.tap_scale, lane, 1, t1h, t1l);
    const bool full_band = r0 + 16 <= rows;
    const int lr = lane >> 3, lc = 4 * (lane & 7);  // loader: rows lr and lr + 8 of the band, 4 columns from lc of the group
    const float* const lrow0 = p.in + (size_t)min(r0 + lr, rows - 1) * p.nx;
    const float* const lrow1 = p.in + (size_t)min(r0 + lr + 8, rows - 1) * p.nx;
    struct Pre {
        f4 a, b;  // rows lr, lr + 8
    };
    auto load = [&](int col0) {
        Pre v;
        const int col = col0 + lc;
        if (col0 >= 0 && col0 + 32 <= p.nx) {
            v.a = *reinterpret_cast<const f4*>(lrow0 + col);
            v.b = *reinterpret_cast<const f4*>(lrow1 + col);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cx = reflect_index(col + e, p.nx);
                v.a[e] = lrow0[cx];
                v.b[e] = lrow1[cx];
            }
        }
        return v;
    };
    // the 32 columns from `col0` (a multiple of 32) into the ring at `slot`; c_old: the references of their slab for
    // the lane's two rows as the table has them (read ahead of time; not used when the slab starts here)
    auto stage = [&](const Pre& v, int col0, int slot, float c_old0, float c_old1, bool& bad, bool& ref_bad) {
        float x0[4], x1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool w0 = wild(v.a[e]), w1 = wild(v.b[e]);
            bad |= w0 | w1;
            x0[e] = w0 ? 0.0f : v.a[e];
            x1[e] = w1 ? 0.0f : v.b[e];
        }
        float c0 = c_old0, c1 = c_old1;
        if ((col0 & 63) == 0) {  // (wave-uniform) the slab's first column: lane & 7 == 0 of every row holds its reference
            ref_bad |= (lane & 7) == 0 && (wild(v.a[0]) || wild(v.b[0]));
            c0 = octet_first(x0[0], lane);
            c1 = octet_first(x1[0], lane);
            if ((lane & 7) == 0) {
                ctab[(floor_div64(col0) & 7) * 16 + lr] = c0;
                ctab[(floor_div64(col0) & 7) * 16 + lr + 8] = c1;
            }
        }
        f16x4 dh, dl;
        split4(x0, -0.25f * c0, dh, dl);
        *reinterpret_cast<f16x4*>(hi + lr * PITCH + slot + lc) = dh;
        *reinterpret_cast<f16x4*>(lo + lr * PITCH + slot + lc) = dl;
        split4(x1, -0.25f * c1, dh, dl);
        *reinterpret_cast<f16x4*>(hi + (lr + 8) * PITCH + slot + lc) = dh;
        *reinterpret_cast<f16x4*>(lo + (lr + 8) * PITCH + slot + lc) = dl;
    };
    // The bands do not walk the row in step: band b's tile order is turned by turn_b tiles (tile = (position + turn_b) mod
    // ntile), so its runs start somewhere else than its neighbours' and the one run that meets the end of the row goes
    // on at its start (a second run-in for that run).  In step, all bands of the DEM sit on the same columns at the
    // same time, 2^17 bytes apart on a 32768-wide DEM, and whether HBM's channels then collide depends on the physical
    // pages: the + 30 ... 45 % slow mode most processes show on some boxes (profiles/r04_pitch_spread.txt; axis 1 alone at
    // sigma 13 / 20: 1.66 / 1.86 ms or 2.40 / 2.43 ms in step, 1.79 / 1.88 ms turned).
    const int run_len = t_last - t_first;
    const int turn = p.group0 != 0 && ntile >= 64 ? (int)(((unsigned)band * 40503u) % (unsigned)ntile) : 0;
    const int run_first = (t_first + turn) % ntile;
    for (int part = 0; part < 2; ++part) {
    const int t_first = part == 0 ? run_first : 0;
    const int t_last = part == 0 ? min(run_first + run_len, ntile) : run_first + run_len - ntile;
    if (t_first >= t_last) continue;
    unsigned ref_wild = 0;  // slabs (slot bits) whose reference column holds a sample that is not a plain finite one
    int last_wild = kNoWild;
    const int x_start = t_first * 32 - Rp;
    {
        ctab[lane] = 0.0f;
        ctab[64 + lane] = 0.0f;
        __builtin_amdgcn_wave_barrier();
        // the references of the slabs the first window holds: the first one may start left of the window (the others are
        // written again, with the same value, when their first column is staged)
        unsigned used = 0;
        for (int s = floor_div64(x_start); s <= floor_div64(x_start + RC - 1); ++s) {
            const int cx = reflect_index(64 * s, p.nx);
            const float v0 = lrow0[cx], v1 = lrow1[cx];
            const bool w0 = wild(v0), w1 = wild(v1);
            if ((lane & 7) == 0) {
                ctab[(s & 7) * 16 + lr] = w0 ? 0.0f : v0;
                ctab[(s & 7) * 16 + lr + 8] = w1 ? 0.0f : v1;
            }
            if (__builtin_amdgcn_ballot_w64(w0 | w1)) ref_wild |= 1u << (s & 7);
            used |= 1u << (s & 7);
        }
        __builtin_amdgcn_wave_barrier();
        bool bad = false, ref_bad = false;  // (ref_bad: covered by the loop above)
        for (int k0 = 0; k0 < RC; k0 += 96) {  // three groups in flight
            Pre v[3];
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (k0 + 32 * u < RC) v[u] = load(x_start + k0 + 32 * u);
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (k0 + 32 * u < RC) {
                    const int col0 = x_start + k0 + 32 * u, sl = (floor_div64(col0) & 7) * 16;
                    stage(v[u], col0, k0 + 32 * u, ctab[sl + lr], ctab[sl + lr + 8], bad, ref_bad);
                }
        }
        if (__builtin_amdgcn_ballot_w64(bad) || (ref_wild & used)) last_wild = t_first + kStay - 1;
    }
    int base = 0;  // ring column of input column x0 - Rp
    const int n = lane & 15, kg = lane >> 4;
    const unsigned out_lane_off = (unsigned)(4 * kg * p.nx + n) * 4u;  // D: column n, rows 4 kg + v
    const _Float16* const ah = hi + n * PITCH + 8 * kg;  // A: row n of the band, 8 window positions from 8 kg
    const _Float16* const al = lo + n * PITCH + 8 * kg;
    // The columns tile t + 1 adds are loaded while tile t - 1 is computed and staged behind tile t: two register sets
    // that swap roles from tile to tile (the loop below takes two tiles per turn, so nothing is copied and nothing
    // waits for a load before its data is due).
    Pre pre_a, pre_b;
    if (t_first + 1 < t_last) pre_a = load(x_start + RC);
    auto tile = [&](const int t, const Pre& pre, Pre& pre_next) {
        const int x0 = t * 32;
        const bool more = t + 1 < t_last;
        const int n0 = x0 - Rp + RC;  // the 32 columns the next tile adds
        if (t + 2 < t_last) pre_next = load(n0 + 32);
        float c_n0 = 0.0f, c_n1 = 0.0f;
        if (more) {
            const int sl = (floor_div64(n0) & 7) * 16;
            c_n0 = ctab[sl + lr];
            c_n1 = ctab[sl + lr + 8];
        }
        // the correction (ahead of the products: nothing waits on LDS behind them): the other slabs' references against
        // the tile's own.  D: column n, rows 4 kg + v
        const int b = x0 >> 6, o = (x0 & 63) + n;
        const f4 cb = *reinterpret_cast<const f4*>(ctab + (b & 7) * 16 + 4 * kg);
        f4 corr0 = {0.0f, 0.0f, 0.0f, 0.0f}, corr1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int d = -NSIDE; d <= NSIDE; ++d) {
            if (d == 0) continue;
            const float w0 = wlds[(d + 2) * 64 + o], w1 = wlds[(d + 2) * 64 + o + 16];
            const f4 cs = *reinterpret_cast<const f4*>(ctab + ((b + d) & 7) * 16 + 4 * kg);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dlt = cs[e] - cb[e];
                corr0[e] = fmaf(dlt, w0, corr0[e]);
                corr1[e] = fmaf(dlt, w1, corr1[e]);
            }
        }
        f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
        int slot = base;
        f16x8 bh[3], bl[3];
        auto fetch = [&](int k) {
            bh[k] = *reinterpret_cast<const f16x8*>(ah + slot);
            bl[k] = *reinterpret_cast<const f16x8*>(al + slot);
            slot += 32;
            slot = slot >= RC ? slot - RC : slot;
        };
        fetch(0);
        fetch(1);
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            if (s + 2 < NK) fetch((s + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);
            f16_products16_pair<NP>(t0h[s], t0l[s], t1h[s], t1l[s], bh[s % 3], bl[s % 3], acc0, acc1);
        }
        {
            const int ox = x0 + n;
            if (full_band && x0 + 32 <= p.nx) {  // (wave-uniform) every row and both sub-tiles: one base, immediate offsets
                char* ub = reinterpret_cast<char*>(p.out + (size_t)r0 * p.nx + x0) + out_lane_off;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    float* o0 = reinterpret_cast<float*>(ub + (size_t)v * p.nx * 4);
                    o0[0] = fmaf(acc0[v], p.out_scale, corr0[v]) + cb[v];
                    o0[16] = fmaf(acc1[v], p.out_scale, corr1[v]) + cb[v];
                }
            } else {
                float* o0 = p.out + (size_t)(r0 + 4 * kg) * p.nx + ox;
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (full_band || r0 + 4 * kg + v < rows) {
                        if (ox < p.nx) o0[(size_t)v * p.nx] = fmaf(acc0[v], p.out_scale, corr0[v]) + cb[v];
                        if (ox + 16 < p.nx) o0[(size_t)v * p.nx + 16] = fmaf(acc1[v], p.out_scale, corr1[v]) + cb[v];
                    }
            }
        }
        if (lane == 0) p.flags[(size_t)band * ntile + t] = last_wild >= t ? 1 : 0;
        if (more) {
            bool bad = false, ref_bad = false;
            stage(pre, n0, base, c_n0, c_n1, bad, ref_bad);  // over the oldest 32 columns
            if ((n0 & 63) == 0) {  // the slab that starts here takes over the slot of the slab 8 before it
                if (__builtin_amdgcn_ballot_w64(ref_bad)) ref_wild |= 1u << (floor_div64(n0) & 7);
                else ref_wild &= ~(1u << (floor_div64(n0) & 7));
            }
            if (__builtin_amdgcn_ballot_w64(bad) || (ref_wild & (1u << (floor_div64(n0) & 7)))) last_wild = t + kStay;
        }
        base += 32;
        base = base >= RC ? base - RC : base;
    };
    for (int t = t_first; t < t_last; t += 2) {
        tile(t, pre_a, pre_b);
        if (t + 1 < t_last) tile(t + 1, pre_b, pre_a);
    }
    }
}


// Both passes of a short isotropic filter (radius 4 ... 16: 4 steps of 16 taps per 32-output tile) in ONE kernel:
// the intermediate plane (4 B written + 4 B read per pixel out of 16) stays in LDS.  A block owns 4 bands of 32
// rows (global multiples of 32) and marches along x in blocks of 64 columns.  Per step: the block's 160 raw rows
// (its 128 + 16 above and below) x 64 columns come in through a double buffer; every wave forms the axis-0 product
// for its band - the arithmetic of gauss_axis0_f16_kernel<4, ., 1>, tile for tile: same rows, same offset (the
// column's sample 16 rows into the band), same steps - rounds it to float32 like the stored intermediate, and puts
// the 32 x 64 block into its own ring of 96 columns; then it takes the 64-column output tile whose window the block
// completes with the arithmetic of gauss_axis1_f16_kernel<4, ., 2, .>.  An output therefore has the bits of the two
// kernels (tests/test_gpu_parity.py::test_gaussian_fused_is_the_two_passes).  The raw blocks sit 16 columns to the
// right of the tile grid, so block u completes the window of tile u; columns beyond the DEM are the reflected
// columns (the axis-0 result of a reflected column is that column's).  Rows are read 160 / 128 times, not twice.
// A sample that is not a plain finite one is staged as 0 and raises *wild_flag: the launcher has queued the two-pass
// kernels behind this one with run_if = wild_flag, and they (with their repair passes) redo the whole plane then.
// (S = 6, radius 17 ... 32: raw blocks of 32 columns - 192 rows of them, twice, plus four rings of 128 columns are
// 117 KB - two per output tile, two run-in blocks.)
template <int NP, int S, int CW>
__global__ __launch_bounds__(256) void gauss_fused_f16_kernel(GaussArgs p, int tile_first, int ntile_rows, int nseg) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    constexpr int Rp = 8 * (S - 2), RAWR = 128 + 2 * Rp, NS = S + 2, RC = 16 * NS, pitch = RC + 4;
    constexpr int BPT = 64 / CW;                  // raw blocks per 64-column output tile
    constexpr int TPR = CW / 4, RPP = 256 / TPR;  // loader: threads per raw row, rows per pass
    constexpr int NPASS = RAWR / RPP;
    static_assert(RAWR % RPP == 0 && (CW == 32 || CW == 64), "loader geometry");
    const int R = p.radius;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, g = lane >> 5;
    float* const rawbuf = L;  // 2 x [RAWR][CW]
    float* const ring = L + 2 * RAWR * CW + wave * (32 * pitch + 32);
    float* const crow = ring + 32 * pitch;
    f16x8 twh[S], twl[S];
    build_tap_blocks<S>(p.taps, R, p.tap_scale, lane, twh, twl);
    const int y_blk = (tile_first + 4 * (int)blockIdx.y) * 32;  // global row of the block's first band
    const int y_band = y_blk + 32 * wave;
    const bool band_on = 4 * (int)blockIdx.y + wave < ntile_rows;
    const int ntile = (p.nx + 63) / 64;
    const int tper = (ntile + nseg - 1) / nseg;
    const int seg_first = blockIdx.x * tper, seg_last = min(seg_first + tper, ntile);
    if (seg_first >= seg_last) return;
    const int row_lo = max(0, p.in_row0), row_hi = min(p.gny, p.in_row0 + p.in_rows);
    const bool rows_inside = y_blk - Rp >= row_lo && y_blk + 128 + Rp <= row_hi;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lr = threadIdx.x / TPR, lcq = (threadIdx.x % TPR) * 4;
    // raw block u: columns CW u + Rp ... (the blocks sit Rp columns to the right of the tile grid, so the last block
    // of a tile completes its window)
    auto load_raw = [&](int u, f4 (&pre)[NPASS]) {
        const int xb = CW * u + Rp;
        if (rows_inside && xb >= 0 && xb + CW <= p.nx) {
            const float* rbp = p.in + (size_t)(y_blk - Rp + lr - p.in_row0) * p.nx + xb + lcq;
#pragma unroll
            for (int q = 0; q < NPASS; ++q) pre[q] = *reinterpret_cast<const f4*>(rbp + (size_t)(RPP * q) * p.nx);
        } else {
#pragma unroll
            for (int q = 0; q < NPASS; ++q) {
                int gy = reflect_index(y_blk - Rp + RPP * q + lr, p.gny);
                gy = min(max(gy, p.in_row0), p.in_row0 + p.in_rows - 1);
                const float* rowp = p.in + (size_t)(gy - p.in_row0) * p.nx;
#pragma unroll
                for (int e = 0; e < 4; ++e) pre[q][e] = rowp[reflect_index(xb + lcq + e, p.nx)];
            }
        }
    };
    auto store_raw = [&](f4 (&pre)[NPASS], float* dst) {
        bool bad = false;
#pragma unroll
        for (int q = 0; q < NPASS; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) bad |= wild(pre[q][e]);
        if (__builtin_amdgcn_ballot_w64(bad)) {
#pragma unroll
            for (int q = 0; q < NPASS; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) pre[q][e] = wild(pre[q][e]) ? 0.0f : pre[q][e];
            if (lane == 0) {
                *p.wild_flag = 1;
                if (p.wild_host) __hip_atomic_store(p.wild_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
#pragma unroll
        for (int q = 0; q < NPASS; ++q) *reinterpret_cast<f4*>(dst + (RPP * q + lr) * CW + lcq) = pre[q];
    };
    const f32x2 quarter = {0.25f, 0.25f};
    const unsigned out_lane_off = (unsigned)(4 * g * p.nx + i) * 4u;
    const bool full_band = y_band >= p.out_row0 && y_band + 32 <= p.out_row0 + p.out_rows;
    // The row blocks do not walk the DEM in step (gauss_axis1_s1_kernel has the reasons): row block y's tile order is turned
    // by a number of tiles of its own, and the run that meets the end of the row goes on at its start (a second run-in).
    const int turn = p.group0 != 0 && ntile >= 32 ? (int)(((unsigned)blockIdx.y * 40503u) % (unsigned)ntile) : 0;
    const int run_first = (seg_first + turn) % ntile, run_len = seg_last - seg_first;
    for (int part = 0; part < 2; ++part) {
    const int t_first = part == 0 ? run_first : 0;
    const int t_last = part == 0 ? min(run_first + run_len, ntile) : run_first + run_len - ntile;
    if (t_first >= t_last) continue;
    const int u0 = BPT * t_first;                       // first block of the first tile
    const int u_start = u0 - (2 * Rp + CW - 1) / CW;    // run-in: the 2 Rp columns left of it
    const int u_last = BPT * t_last - 1;
    {
        f4 pre[NPASS];
        load_raw(u_start, pre);
        store_raw(pre, rawbuf);
    }
    __syncthreads();
    int cur = 0;
    int rb = 0;                                   // ring column of the window of the next output tile (global column 64 t - Rp)
    int rel = CW * (u_start - u0) + 2 * Rp;       // column of block u relative to the first tile's window (may start below 0)
    for (int u = u_start; u <= u_last; ++u) {
        f4 pre[NPASS];
        const bool more = u < u_last;
        if (more) load_raw(u + 1, pre);
        const bool tile_done = u >= u0 && (u + 1) % BPT == 0;
        if (band_on) {
            // ---- axis 0: the band's 32 rows of raw block u -> ring ----
            const float* raw = rawbuf + cur * (RAWR * CW);
#pragma unroll
            for (int nt = 0; nt < CW / 32; ++nt) {
                if (rel + 32 * nt < 0) continue;  // (run-in of S = 4: only the second half of the block is in the window)
                const int col = 32 * nt + i;
                const float c0 = raw[(32 * wave + Rp + 16) * CW + col];
                const f32x2 mcq = {-0.25f * c0, -0.25f * c0};
                f32x16 acc;
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
                const float* bl = raw + (32 * wave + 8 * g) * CW + col;
                float xa[8], xb[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) xa[q] = bl[q * CW];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    f16x8 dh, dl;
                    // the next step's samples are on their way while this step's are split (as in the two-pass kernels)
                    if (s & 1) {
                        if (s + 1 < S) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) xa[q] = bl[(16 * (s + 1) + q) * CW];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        split8(xb, quarter, mcq, dh, dl);
                    } else {
                        if (s + 1 < S) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) xb[q] = bl[(16 * (s + 1) + q) * CW];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        split8(xa, quarter, mcq, dh, dl);
                    }
                    f16_products<false, NP>(twh[s], twl[s], dh, dl, acc);
                }
                int wc = (rel + 32 * nt) % RC;
#pragma unroll
                for (int v = 0; v < 16; ++v) ring[((v & 3) + 8 * (v >> 2) + 4 * g) * pitch + wc + i] = fmaf(acc[v], p.out_scale, c0);
            }
            // ---- axis 1: the output tile this block completes, columns 64 t ... 64 t + 63 ----
            if (tile_done) {
                const int x0 = 64 * ((u + 1) / BPT - 1);
                int sc = rb + Rp + 32;
                sc = sc >= RC ? sc - RC : sc;
                const float c = ring[i * pitch + sc];
                if (g == 0) crow[i] = c;
                const f32x2 mcq = {-0.25f * c, -0.25f * c};
                f32x16 acc[2];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[m][v] = 0.0f;
                const float* const al = ring + i * pitch + 8 * g;
                int slot = rb;
                auto fetch = [&](float (&x)[8]) {
                    const f4 lo4 = *reinterpret_cast<const f4*>(al + slot), hi4 = *reinterpret_cast<const f4*>(al + slot + 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        x[q] = lo4[q];
                        x[4 + q] = hi4[q];
                    }
                    slot += 16;
                    slot = slot >= RC ? slot - RC : slot;
                };
                float xa[8], xb[8];
                fetch(xa);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    f16x8 dh, dl;
                    if (s & 1) {
                        if (s + 1 < NS) fetch(xa);
                        __builtin_amdgcn_sched_barrier(0);
                        split8(xb, quarter, mcq, dh, dl);
                    } else {
                        if (s + 1 < NS) fetch(xb);
                        __builtin_amdgcn_sched_barrier(0);
                        split8(xa, quarter, mcq, dh, dl);
                    }
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        if (s - 2 * m >= 0 && s - 2 * m < S) f16_products<true, NP>(twh[s - 2 * m], twl[s - 2 * m], dh, dl, acc[m]);
                }
                float cr[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) cr[v] = crow[(v & 3) + 8 * (v >> 2) + 4 * g];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int xm = x0 + 32 * m, ox = xm + i;
                    if (ox < p.nx) {
                        if (full_band) {
                            char* ub = reinterpret_cast<char*>(p.out + (size_t)(y_band - p.out_row0) * p.nx + xm);
#pragma unroll
                            for (int v = 0; v < 16; ++v)
                                *reinterpret_cast<float*>(ub + (size_t)((v & 3) + 8 * (v >> 2)) * p.nx * 4 + out_lane_off) = fmaf(acc[m][v], p.out_scale, cr[v]);
                        } else {
#pragma unroll
                            for (int v = 0; v < 16; ++v) {
                                const int oy = y_band + (v & 3) + 8 * (v >> 2) + 4 * g;
                                if (oy >= p.out_row0 && oy < p.out_row0 + p.out_rows) p.out[(size_t)(oy - p.out_row0) * p.nx + ox] = fmaf(acc[m][v], p.out_scale, cr[v]);
                            }
                        }
                    }
                }
            }
        }
        if (tile_done) {
            rb += 64;
            rb = rb >= RC ? rb - RC : rb;
        }
        rel += CW;
        if (rel >= 64 * RC) rel -= 63 * RC;  // (kept small; only rel % RC and its sign before the first tile matter)
        if (more) store_raw(pre, rawbuf + (cur ^ 1) * (RAWR * CW));
        __syncthreads();
        cur ^= 1;
    }
    }
}

// Repair pass of the f16 route: the outputs of a marked tile whose own window holds a sample that is not a plain
// finite one (non-finite, or |x| > 1e5) are taken tap by tap in float32 - an ascending chain of fused multiply-adds
// on x - c, c the tile's offset - so they are NaN / inf exactly where the reference's are; every other output
// keeps what the matrix cores gave it.
template <bool AXIS1>
__global__ __launch_bounds__(64) void gauss_f16_repair_kernel(GaussArgs p, int units_a, int units_b, int first_a, int rows_plane) {
    if (p.run_if && *p.run_if == 0) return;
    const int lane = threadIdx.x;
    const long units = (long)units_a * units_b;
    const int R = p.radius;
    // A lane takes 16 consecutive outputs along the filter axis.  `at(m)`: the sample m places after the first
    // output's window start (so output k owns at(k) ... at(k + 2 R), its own sample is at(k + R)).  The marks are
    // conservative and a marked tile is mostly clean windows, or mostly NaN (a sea): the lane first COUNTS the samples
    // of the first window that are not plain finite ones (loads 8 at a time, nothing depends on them), then slides
    // the count from output to output (one sample in, one out) and runs the float32 chain - left at the first NaN -
    // only where the count is not 0.  (The first version ran 16 full chains of dependent loads per lane: 8192^2 with
    // a NaN third at sigma 30.25 took 135 ms instead of 0.45; now ~1 ms.)
    auto sixteen = [&](auto at, float c, auto dst, auto wanted) {
        // (deep in a NaN sea every output is NaN by its own sample: 16 loads in flight settle the lane)
        float own[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) own[k] = at(k + R);
        bool all_nan = true;
#pragma unroll
        for (int k = 0; k < 16; ++k) all_nan = all_nan && __builtin_isnan(own[k]);
        if (all_nan) {
            for (int k = 0; k < 16; ++k)
                if (wanted(k)) *dst(k) = __builtin_nanf("");
            return;
        }
        int cnt = 0;
        for (int m0 = 0; m0 <= 2 * R; m0 += 8) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = m0 + e <= 2 * R ? at(m0 + e) : 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) cnt += wild(v[e]) ? 1 : 0;
        }
        for (int k = 0; k < 16; ++k) {
            if (cnt != 0 && wanted(k)) {
                float acc = 0.0f;
                bool nan = false;
                for (int q = 0; q <= 2 * R; ++q) {
                    const float v = at(k + q);
                    if (__builtin_isnan(v)) {
                        nan = true;
                        break;
                    }
                    acc = fmaf(p.taps[q], v - c, acc);
                }
                *dst(k) = nan ? __builtin_nanf("") : c + acc;
            }
            if (k < 15) cnt += (wild(at(k + 2 * R + 1)) ? 1 : 0) - (wild(at(k)) ? 1 : 0);
        }
    };
    for (long base = (long)blockIdx.x * 64; base < units; base += (long)gridDim.x * 64) {
        const long mine = base + lane;
        bool mark = false;
        if (mine < units) {
            if (p.fine_cols == 0) {
                mark = p.flags[mine] != 0;
            } else {  // the two 16-row bands of the unit
                const int fa = (int)(mine / units_b) * 2, fb = (int)(mine % units_b);
                mark = p.flags[(size_t)fa * p.fine_cols + fb] != 0 || (fa + 1 < p.fine_rows && p.flags[(size_t)(fa + 1) * p.fine_cols + fb] != 0);
            }
        }
        unsigned long long marked = __builtin_amdgcn_ballot_w64(mark);
        while (marked) {
            const int bit = __builtin_ctzll(marked);
            marked &= marked - 1;
            const long u = base + bit;
            const int a = (int)(u / units_b), b = (int)(u % units_b);
            const int j = lane & 31, h = lane >> 5;
            if (!AXIS1) {
                const int y0 = (first_a + a) * 32, x = 32 * b + j;
                if (x >= p.nx) continue;
                auto in_at = [&](int gy) {
                    gy = reflect_index(gy, p.gny);
                    gy = min(max(gy, p.in_row0), p.in_row0 + p.in_rows - 1);
                    return p.in[(size_t)(gy - p.in_row0) * p.nx + x];
                };
                float c = in_at(y0 + 16);
                c = wild(c) ? 0.0f : c;
                const int oy0 = y0 + 16 * h;  // first output row of this lane
                const int row_lo = max(0, p.in_row0), row_hi = min(p.gny, p.in_row0 + p.in_rows);
                auto dst = [&](int k) { return p.out + (size_t)(oy0 + k - p.out_row0) * p.nx + x; };
                auto wanted = [&](int k) { return oy0 + k >= p.out_row0 && oy0 + k < p.out_row0 + p.out_rows; };
                if (oy0 - R >= row_lo && oy0 + 16 + R < row_hi) {  // every window inside the block: a pointer and a stride
                    const float* w0 = p.in + (size_t)(oy0 - R - p.in_row0) * p.nx + x;
                    const size_t nx = (size_t)p.nx;
                    sixteen([&](int m) { return w0[(size_t)m * nx]; }, c, dst, wanted);
                } else {
                    sixteen([&](int m) { return in_at(oy0 - R + m); }, c, dst, wanted);
                }
            } else {
                const int r = 32 * a + j, x0 = 32 * b + 16 * h;  // first output column of this lane
                if (r >= rows_plane) continue;
                const float* row = p.in + (size_t)r * p.nx;
                float c = row[reflect_index(32 * b + 16, p.nx)];
                c = wild(c) ? 0.0f : c;
                auto dst = [&](int k) { return p.out + (size_t)r * p.nx + x0 + k; };
                auto wanted = [&](int k) { return x0 + k < p.nx; };
                if (x0 - R >= 0 && x0 + 16 + R < p.nx) {
                    const float* w0 = row + x0 - R;
                    sixteen([&](int m) { return w0[m]; }, c, dst, wanted);
                } else {
                    sixteen([&](int m) { return row[reflect_index(x0 - R + m, p.nx)]; }, c, dst, wanted);
                }
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------------------
// repair passes: one wave per block; enough blocks to fill the chip when many tiles are marked (a DEM with a NaN sea),
// a short scan of the marks when none is
constexpr long kRepairBlocks = 8192;
int upload_weights(int slot, double sigma, int kb, GaussArgs* a) {
    const int R = gaussian_radius(sigma);
    std::vector<double> w(2 * R + 1);
    double sum = 0.0;
    for (int k = -R; k <= R; ++k) {
        w[k + R] = std::exp(-0.5 / (sigma * sigma) * (double)k * (double)k);
        sum += w[k + R];
    }
    const int nchunks = (2 * R + 1 + kb - 1) / kb;
    std::vector<float> padded((size_t)nchunks * kb, 0.0f);
    for (int k = 0; k <= 2 * R; ++k) padded[k] = (float)(w[k] / sum);
    // behind the plain taps (kb == 1): the slab table of the split-once kernels, W[d + 2][o] = the sum of the (float32)
    // taps that the output at position o = 0 ... 63 of its slab lays over the slab d slabs further on
    const size_t ntaps = padded.size();
    if (kb == 1) {
        padded.resize(ntaps + 5 * 64, 0.0f);
        for (int d = -2; d <= 2; ++d)
            for (int o = 0; o < 64; ++o) {
                double acc = 0.0;
                for (int k = 64 * d; k < 64 * d + 64; ++k) {
                    const int q = k - o + R;
                    if (q >= 0 && q <= 2 * R) acc += (double)padded[q];
                }
                padded[ntaps + (size_t)(d + 2) * 64 + o] = (float)acc;
            }
    }
    void* d = nullptr;
    TOPO_TRY(upload_table(slot, padded.data(), padded.size() * sizeof(float), &d));
    a->taps = (const float*)d;
    a->wtab = kb == 1 ? (const float*)d + ntaps : nullptr;
    a->radius = R;
    a->nchunks = nchunks;
    return TOPO_AMD_OK;
}

// wide tiling for long filters, narrow for short ones (fewer padded taps)
bool wide_tiling(int radius) {
    constexpr int from = 24;
    return radius >= from;
}

// The MFMA kernels take the filters whose ring fits LDS (radius <= 121, sigma <= 30.3).  Same-box sweeps on the
// 32768^2 bench DEM (tools/grad_time.py, profiles/r02_gauss_mfma.txt).  Gaussian alone: they win at every radius
// tried, down to 4 (sigma 1.1 ... 1.75: 3.5-3.6 ms against 5.6 ms for the vector-ALU pair; sigma 2 ... 3.75: 3.9-4.2
// against 6.0-6.5; sigma 30.25: 11.3 against 23.0).  Gradient: the vector-ALU axis-1 kernel has the epilogue fused
// in while the MFMA route pays a separate epilogue (36 instead of 28 B/pixel of HBM traffic, ~8.3 ms whatever the
// radius); with the final kernels and the epilogue overlapped by row chunks the MFMA route wins from radius 8
// (sigma 2 ... 3.75: 8.3-8.4 against 8.7-10.5 ms, the fused kernel being bimodal from box to box; sigma 4: 7.85
// against 9.26; sigma 7: 8.2 against 9.8) and loses below (sigma 1.1 ... 1.75: 8.3 against 7.65).
// Below radius 16 the kernels need more of the caller: the per-column accumulation offset of an axis-0 tile is the
// sample 16 rows into the (global) 32-row tile, and a row block holds that row for every tile it computes only
// when it carries 16 ghost rows (17 for the gradient) instead of R (R + 1).  topo_amd_halo_rows asks for them;
// a block that comes with less takes the vector-ALU kernels (mfma_rows_ok), whose results differ from the
// matrix-core ones in the last bits: row-block bit-identity holds for ghost depths of topo_amd_halo_rows or more
// (the pre-smoothing of topo.tpi / topo.std included).  The two smooths of an anisotropic gradient keep the
// radius-16 floor: they have two radii and one ghost depth.
constexpr int kMfmaSmallFloor = 4;
int mfma_min_radius_impl(bool for_gradient) {
    static const int from_gauss = [] {
        const char* e = std::getenv("TOPO_AMD_GAUSS_MFMA_MIN_RADIUS");
        return std::max(kMfmaSmallFloor, e && *e ? std::atoi(e) : 4);
    }();
    static const int from_grad = from_gauss;
    // (round 3: 4, was 8: with both passes in one kernel the route is 6.1 ms at sigma 1.25 on 32768^2, the
    // vector-ALU kernel with the epilogue fused in 7.4)
    return for_gradient ? from_grad : from_gauss;
}

// set by launch_gaussian / launch_gradient for the call: the block's samples are mostly beyond the f16 kernels' range
// (dem_memo_mostly_large): vector-ALU kernels
thread_local bool t_no_mfma = false;
bool mfma_radius(int R, int nx, bool for_gradient = false, bool small_ok = true) {
    if (t_no_mfma) return false;
    // (any width from 4 columns: the loaders' 16-byte loads only need dword alignment, which a row of any length has)
    return R >= std::max(small_ok ? kMfmaSmallFloor : 16, mfma_min_radius_impl(for_gradient)) && R <= 121 && nx >= 4;
}

// the accumulation-offset row of every axis-0 tile that holds one of the block's output rows is inside the block
// (or the block reaches the DEM edge there)
bool mfma_rows_ok(const Block& b, int R) {
    if (R >= 16) return true;
    const int y_first = b.out_row0 / 32 * 32 + 16, y_last = (b.out_row0 + b.out_rows - 1) / 32 * 32 + 16;
    const bool top = b.in_row0 <= 0 || y_first >= b.in_row0;
    const bool bottom = b.in_row0 + b.in_rows >= b.gny || y_last <= b.in_row0 + b.in_rows - 1;
    return top && bottom;
}

int upload_plain_weights(int slot, double sigma, GaussArgs* a) {
    return upload_weights(slot, sigma, 1, a);  // 2 R + 1 taps, no padding
}

// ---- f16 route: scales, step count, dispatch over the step count ----
#ifndef TOPO_F16_NP
#define TOPO_F16_NP 3  // products per tap block: th h + tm h + th l (4 adds tm l: 7 % slower, same error; lab builds)
#endif
int f16_steps(int R);
thread_local bool t_axis0_one_tile = false;  // set around the two-pass launches queued behind a fused launch
// two MFMA tiles per offset (MT = 2): axis 1 always; axis 0 from radius 32 (the offset row is 32 rows into a
// 64-row tile).
int f16_mt(bool axis1, int R) {
    if (axis1) return f16_steps(R) < 18 ? 2 : 1;  // the widest window fits LDS 3 times with MT = 2: 2.95 ms against 2.66 with MT = 1
    // (behind a fused launch - radius 32 ... 47 of the gradient - the two-pass kernels must give the fused kernel's bits,
    // and its bands are 32-row tiles)
    if (t_axis0_one_tile) return 1;
    return R >= 32 ? 2 : 1;
}
int f16_steps(int R) { return 2 + (R + 15) / 16 * 2; }  // 16-sample steps of a 32-output tile's window: 32 + 2 Rp, Rp = R rounded up to 16
void set_f16_scales(double sigma, GaussArgs* a) {
    const int R = gaussian_radius(sigma);
    double sum = 0.0;
    for (int k = -R; k <= R; ++k) sum += std::exp(-0.5 / (sigma * sigma) * (double)k * (double)k);
    const float peak = (float)(1.0 / sum);  // the largest tap, as upload_weights rounds it
    const int e = 9 - std::ilogb(peak);     // peak 2^e in [512, 1024)
    a->tap_scale = std::ldexp(1.0f, e);
    a->out_scale = std::ldexp(1.0f, 2 - e);  // the data goes in divided by 4
}
constexpr int f16_axis1_waves(int S, int MT) { return S == 18 && MT == 2 ? 3 : 4; }  // 4 x 32 rows x 324 floats do not fit 160 KB
template <int S, int MT>
int launch_f16_axis0(dim3 grid, const GaussArgs& a, int tile_first, int ntiles, int per) {
    Context& c = ctx();
    static bool ready = false;
    if (!ready) {
        TOPO_HIP(hipFuncSetAttribute((const void*)gauss_axis0_f16_kernel<S, TOPO_F16_NP, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ready = true;
    }
    const size_t lds = (size_t)(16 * (S + 2 * (MT - 1))) * kMfmaCols * sizeof(float);
    hipLaunchKernelGGL((gauss_axis0_f16_kernel<S, TOPO_F16_NP, MT>), grid, dim3(256), lds, c.compute, a, tile_first, ntiles, per);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}
template <int S, int MT>
int launch_f16_axis1(long waves, const GaussArgs& a, int rows, int nseg) {
    Context& c = ctx();
    constexpr int NW = f16_axis1_waves(S, MT);
    static bool ready = false;
    if (!ready) {
        TOPO_HIP(hipFuncSetAttribute((const void*)gauss_axis1_f16_kernel<S, TOPO_F16_NP, MT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ready = true;
    }
    const size_t lds = NW * (size_t)(32 * (16 * (S + 2 * (MT - 1)) + 4) + 32) * sizeof(float);
    hipLaunchKernelGGL((gauss_axis1_f16_kernel<S, TOPO_F16_NP, MT, NW>), dim3((unsigned)((waves + NW - 1) / NW)), dim3(64 * NW), lds, c.compute, a, rows, nseg);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}
// (every band / row block turns its tile order by a number of tiles of its own: profiles/r04_pitch_spread.txt)
constexpr bool gauss_turn() { return true; }
// split once (gauss_axis*_s1_kernel): radius 49 ... 121.  TOPO_AMD_GAUSS_SPLIT_ONCE=0: the tile kernels everywhere (A/B)
bool split_once(int steps) {
    static const bool on = [] {
        const char* e = std::getenv("TOPO_AMD_GAUSS_SPLIT_ONCE");
        return !(e && *e == '0');
    }();
    return on && steps >= 10;
}
// NK steps of 32 window positions per 16-output tile: Rp = 16 (NK - 1) >= R
// (Rp a multiple of 32: the groups of 32 columns the march stages then never straddle a slab)
int s1_steps(int R) { return (R + 31) / 32 * 2 + 1; }
template <int NK>
int launch_s1_axis1(long waves, const GaussArgs& a, int rows, int nseg) {
    Context& c = ctx();
    static bool ready = false;
    if (!ready) {
        TOPO_HIP(hipFuncSetAttribute((const void*)gauss_axis1_s1_kernel<NK, TOPO_F16_NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ready = true;
    }
    const size_t lds = (8 * (size_t)(16 * (32 * NK + kS1Pad) + 8 * 16) + 5 * 64) * sizeof(float);
    static_assert((8 * (size_t)(16 * (32 * 9 + kS1Pad) + 8 * 16) + 5 * 64) * sizeof(float) <= 160 * 1024, "the widest ring fits LDS");
    hipLaunchKernelGGL((gauss_axis1_s1_kernel<NK, TOPO_F16_NP>), dim3((unsigned)((waves + 7) / 8)), dim3(512), lds, c.compute, a, rows, nseg);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}
int launch_s1_axis1_any(int nk, long waves, const GaussArgs& a, int rows, int nseg) {
    switch (nk) {
        case 5: return launch_s1_axis1<5>(waves, a, rows, nseg);
        case 7: return launch_s1_axis1<7>(waves, a, rows, nseg);
        case 9: return launch_s1_axis1<9>(waves, a, rows, nseg);
    }
    set_error("gaussian (split-once matrix-core route): no kernel for this radius");
    return TOPO_AMD_EUNSUP;
}
#define TOPO_F16_STEPS(X) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18)
int launch_f16_axis0_any(int steps, int mt, dim3 grid, const GaussArgs& a, int tile_first, int ntiles, int per) {
    switch (steps) {
#define TOPO_F16_CASE(S_) \
    case S_: return mt == 2 ? launch_f16_axis0<S_, 2>(grid, a, tile_first, ntiles, per) : launch_f16_axis0<S_, 1>(grid, a, tile_first, ntiles, per);
        TOPO_F16_STEPS(TOPO_F16_CASE)
#undef TOPO_F16_CASE
    }
    set_error("gaussian (f16 matrix-core route): no kernel for this radius");
    return TOPO_AMD_EUNSUP;
}
int launch_f16_axis1_any(int steps, int mt, long waves, const GaussArgs& a, int rows, int nseg) {
    switch (steps) {
#define TOPO_F16_CASE(S_) \
    case S_: return mt == 2 ? launch_f16_axis1<S_, 2>(waves, a, rows, nseg) : launch_f16_axis1<S_, 1>(waves, a, rows, nseg);
        TOPO_F16_STEPS(TOPO_F16_CASE)
#undef TOPO_F16_CASE
    }
    set_error("gaussian (f16 matrix-core route): no kernel for this radius");
    return TOPO_AMD_EUNSUP;
}

// set around the two-pass launches that follow a fused launch: they run only if the fused kernel raised its flag
thread_local const int* t_run_if = nullptr;

// axis 0 on the f16 route: tiles of 32 MT rows on the global grid, one flag byte per 32 x 32 unit
int run_axis0_f16(const Block& b, GaussArgs a, double sigma) {
    Context& c = ctx();
    set_f16_scales(sigma, &a);
    a.run_if = t_run_if;
    a.wild_flag = nullptr;
    const int mt = f16_mt(false, a.radius), tile = 32 * mt;
    const int tile_first = b.out_row0 / tile;
    const int ntiles = (b.out_row0 + b.out_rows - 1) / tile - tile_first + 1;
    const int strips = (b.nx + kMfmaCols - 1) / kMfmaCols;
    // blocks the chip holds at once: the ring of the long filters takes a CU's LDS (one block per CU), two fit up to 80 KB.
    // (Round 6: this was 2 x the CUs whatever the ring - at radius 49 ... 121 on a 16384-column raster 512 blocks of one per CU:
    // two rounds, each run restaging its 2 Rp halo rows for half as many tiles.)
    const size_t ring_bytes = (size_t)(16 * (f16_steps(a.radius) + 2 * (mt - 1))) * kMfmaCols * sizeof(float);
    const int per_cu = ring_bytes <= 80 * 1024 ? 2 : 1;
    int splits = (per_cu * c.num_cu + strips - 1) / strips;
    const int min_run = 8 / mt;  // a cut restages 2 Rp rows: runs of 256 rows or more
    splits = std::max(1, std::min(splits, ntiles / min_run > 0 ? ntiles / min_run : 1));
    if (strips >= c.num_cu) splits = 1;
    const int per = (ntiles + splits - 1) / splits;
    TOPO_TRY(check_grid_rows((ntiles + per - 1) / per, "gaussian (matrix-core axis 0)"));
    dim3 grid(strips, (ntiles + per - 1) / per);
    const long units = (long)ntiles * mt * strips * 4;
    void* flags = nullptr;
    TOPO_TRY(workspace(10, (size_t)units, &flags));
    a.flags = (unsigned char*)flags;
    TOPO_TRY(launch_f16_axis0_any(f16_steps(a.radius), mt, grid, a, tile_first, ntiles, per));
    hipLaunchKernelGGL(gauss_f16_repair_kernel<false>, dim3((unsigned)std::min<long>(kRepairBlocks, (units + 63) / 64)), dim3(64), 0, c.compute,
                       a, ntiles * mt, strips * 4, tile_first * mt, 0);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

// axis 1 on the split-once route: bands of 16 rows, tiles of 32 columns, one flag byte per tile
int run_axis1_s1(GaussArgs a, int rows, int nx) {
    Context& c = ctx();
    const int nk = s1_steps(a.radius);
    const int bands = (rows + 15) / 16, ntile = (nx + 31) / 32;
    // one block of 8 waves per CU: the cut of a band's tiles into runs with the least rounds x (tiles per run + run-in),
    // runs no shorter than 512 columns
    const long slots = 8L * c.num_cu;
    const int halo_tiles = nk;
    int nseg = 1;
    long best = -1;
    for (int n = 1; n <= std::max(1, ntile / 16); ++n) {
        const long rounds = ((long)bands * n + slots - 1) / slots;
        const long cost = rounds * ((ntile + n - 1) / n + halo_tiles);
        if (best < 0 || cost < best) {
            best = cost;
            nseg = n;
        }
    }
    const long waves = (long)bands * nseg;
    void* flags = nullptr;
    TOPO_TRY(workspace(10, (size_t)bands * ntile, &flags));
    a.flags = (unsigned char*)flags;
    a.fine_rows = bands;
    a.fine_cols = ntile;
    a.group0 = gauss_turn() ? 1 : 0;
    TOPO_TRY(launch_s1_axis1_any(nk, waves, a, rows, nseg));
    const int units_a = (rows + 31) / 32, units_b = (nx + 31) / 32;
    const long units = (long)units_a * units_b;
    hipLaunchKernelGGL(gauss_f16_repair_kernel<true>, dim3((unsigned)std::min<long>(kRepairBlocks, (units + 63) / 64)), dim3(64), 0, c.compute,
                       a, units_a, units_b, 0, rows);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

int run_axis1_f16(GaussArgs a, int rows, int nx, double sigma) {
    Context& c = ctx();
    set_f16_scales(sigma, &a);
    a.run_if = t_run_if;
    a.wild_flag = nullptr;
    const int steps = f16_steps(a.radius);
    a.group0 = gauss_turn() ? 1 : 0;
    if (split_once(steps)) return run_axis1_s1(a, rows, nx);
    const int mt = f16_mt(true, a.radius);
    const int nw = steps == 18 && mt == 2 ? 3 : 4;
    const int bands = (rows + 31) / 32;
    const int ntile = (nx + 32 * mt - 1) / (32 * mt), nunit = (nx + 31) / 32;
    // one block of nw waves per CU (the rings take the LDS): pick the cut of a band's tiles into runs with the least
    // rounds x (tiles per run + the halo a run restages), runs no shorter than 512 columns
    const long slots = (long)nw * c.num_cu;
    const int halo_tiles = (2 * a.radius + 32 * mt - 1) / (32 * mt);
    int nseg = 1;
    long best = -1;
    for (int n = 1; n <= std::max(1, ntile / (16 / mt)); ++n) {
        const long rounds = ((long)bands * n + slots - 1) / slots;
        const long cost = rounds * ((ntile + n - 1) / n + halo_tiles);
        if (best < 0 || cost < best) {
            best = cost;
            nseg = n;
        }
    }
    const long waves = (long)bands * nseg;
    const long units = (long)bands * nunit;
    void* flags = nullptr;
    TOPO_TRY(workspace(10, (size_t)units, &flags));
    a.flags = (unsigned char*)flags;
    TOPO_TRY(launch_f16_axis1_any(steps, mt, waves, a, rows, nseg));
    hipLaunchKernelGGL(gauss_f16_repair_kernel<true>, dim3((unsigned)std::min<long>(kRepairBlocks, (units + 63) / 64)), dim3(64), 0, c.compute,
                       a, bands, nunit, 0, rows);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

int run_axis0_mfma(const Block& b, double sigma, float* out, int table_slot) {
    GaussArgs a{};
    TOPO_TRY(upload_plain_weights(table_slot, sigma, &a));
    a.in = b.in;
    a.out = out;
    a.in_rows = b.in_rows;
    a.in_row0 = b.in_row0;
    a.gny = b.gny;
    a.nx = b.nx;
    a.out_row0 = b.out_row0;
    a.out_rows = b.out_rows;
    a.group0 = 0;
    return run_axis0_f16(b, a, sigma);
}

int run_axis1_mfma(const float* in, int rows, int nx, double sigma, float* out, int table_slot) {
    GaussArgs a{};
    TOPO_TRY(upload_plain_weights(table_slot, sigma, &a));
    a.in = in;
    a.out = out;
    a.in_rows = rows;
    a.in_row0 = 0;
    a.gny = rows;
    a.nx = nx;
    a.out_row0 = 0;
    a.out_rows = rows;
    a.group0 = 0;
    return run_axis1_f16(a, rows, nx, sigma);
}

// ---- one kernel for both passes (radius 4 ... 16, one sigma) ----
bool fused_radius(int R, bool wide_too) {
    static const bool on = [] {
        const char* e = std::getenv("TOPO_AMD_GAUSS_FUSED");
        return !(e && *e == '0');
    }();
    constexpr int max_r = 47;
    // (the raw blocks and rings of 10 steps no longer fit LDS.)  Radius 32 ... 47 only where the caller says so - the
    // gradient, whose smooth shares HBM with the epilogue (sigma 10: 7.9 -> 6.7 ms); alone the two passes with 64-row
    // tiles on axis 0 are faster there (Gaussian sigma 10: 3.3 ms against 4.4)
    return on && R >= kMfmaSmallFloor && R <= std::min(wide_too ? 47 : 31, max_r);
}
template <int S, int CW>
int launch_fused_f16(dim3 grid, const GaussArgs& a, int tile_first, int ntile_rows, int nseg) {
    Context& c = ctx();
    static bool ready = false;
    if (!ready) {
        TOPO_HIP(hipFuncSetAttribute((const void*)gauss_fused_f16_kernel<TOPO_F16_NP, S, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ready = true;
    }
    constexpr int Rp = 8 * (S - 2);
    const size_t lds = (size_t)(2 * (128 + 2 * Rp) * CW + 4 * (32 * (16 * (S + 2) + 4) + 32)) * sizeof(float);
    hipLaunchKernelGGL((gauss_fused_f16_kernel<TOPO_F16_NP, S, CW>), grid, dim3(256), lds, c.compute, a, tile_first, ntile_rows, nseg);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}
// *flag_out == nullptr on entry: the launch gets a flag of its own, cleared here.  Otherwise it raises the caller's flag
// (the chunked gradient clears one flag per call and queues the two-pass route behind all of its chunks).
int run_fused_f16(const Block& b, double sigma, float* out, int table_slot, const int** flag_out) {
    Context& c = ctx();
    GaussArgs a{};
    TOPO_TRY(upload_plain_weights(table_slot, sigma, &a));
    a.in = b.in;
    a.out = out;
    a.in_rows = b.in_rows;
    a.in_row0 = b.in_row0;
    a.gny = b.gny;
    a.nx = b.nx;
    a.out_row0 = b.out_row0;
    a.out_rows = b.out_rows;
    a.group0 = gauss_turn() ? 1 : 0;
    a.flags = nullptr;
    a.run_if = nullptr;
    set_f16_scales(sigma, &a);
    if (*flag_out == nullptr) {
        void* flag = nullptr;
        TOPO_TRY(workspace(11, 64, &flag));
        TOPO_HIP(hipMemsetAsync(flag, 0, sizeof(int), c.compute));
        *flag_out = (const int*)flag;
    }
    a.wild_flag = const_cast<int*>(*flag_out);
    a.wild_host = dem_memo_wild_word(b);
    const int tile_first = b.out_row0 / 32;
    const int ntile_rows = (b.out_row0 + b.out_rows - 1) / 32 - tile_first + 1;
    const int row_blocks = (ntile_rows + 3) / 4;
    const int ntile = (b.nx + 63) / 64;
    // one block per CU (the LDS): the cut of the columns into runs with the least rounds x (tiles per run + run-in)
    int nseg = 1;
    long best = -1;
    for (int n = 1; n <= std::max(1, ntile / 4); ++n) {
        const long rounds = ((long)row_blocks * n + c.num_cu - 1) / c.num_cu;
        const long cost = rounds * ((ntile + n - 1) / n + 1);
        if (best < 0 || cost < best) {
            best = cost;
            nseg = n;
        }
    }
    TOPO_TRY(check_grid_rows(row_blocks, "gaussian (fused matrix-core kernel)"));
    const dim3 grid(nseg, row_blocks);
    if (f16_steps(a.radius) == 4) return launch_fused_f16<4, 64>(grid, a, tile_first, ntile_rows, nseg);
    if (f16_steps(a.radius) == 6) return launch_fused_f16<6, 32>(grid, a, tile_first, ntile_rows, nseg);
    return launch_fused_f16<8, 32>(grid, a, tile_first, ntile_rows, nseg);
}
// both passes with one sigma on the matrix cores: rows `b.out_row0 ...` -> out; tmp: a plane of the same size
// deferred != nullptr (chunked gradient): on the fused route only the fused kernel is launched, raising *deferred (a flag
// the caller cleared) when it meets a sample that is not a plain finite one; the caller queues smooth_two_pass_if behind
// its last chunk.  Returns whether the two passes are still owed in *owed.
int smooth_both_mfma(const Block& b, double sigma, float* tmp, float* out, int table_slot, bool for_gradient,
                     const int* deferred = nullptr, bool* owed = nullptr) {
    // (a DEM the fused kernel has met a non-finite or huge sample on - dem_memo, common.hpp - goes straight to the two
    // passes: the fused attempt would only be thrown away again)
    // (the two passes behind - or instead of - a fused launch keep its tiling of axis 0, hence its bits)
    const bool fused_class = fused_radius(gaussian_radius(sigma), for_gradient);
    const bool fused = fused_class && !dem_memo_wild(b);
    const int* flag = fused ? deferred : nullptr;
    if (fused) TOPO_TRY(run_fused_f16(b, sigma, out, table_slot, &flag));
    if (owed) *owed = fused && deferred != nullptr;
    if (fused && deferred != nullptr) return TOPO_AMD_OK;
    t_run_if = flag;
    t_axis0_one_tile = fused_class;
    int r = run_axis0_mfma(b, sigma, tmp, table_slot);
    if (r == TOPO_AMD_OK) r = run_axis1_mfma(tmp, b.out_rows, b.nx, sigma, out, table_slot + 1);
    t_run_if = nullptr;
    t_axis0_one_tile = false;
    return r;
}
// the two passes over rows `b.out_row0 ...`, run only if *flag != 0 (queued behind fused launches that share the flag)
int smooth_two_pass_if(const Block& b, double sigma, float* tmp, float* out, int table_slot, const int* flag) {
    t_run_if = flag;
    t_axis0_one_tile = true;
    int r = run_axis0_mfma(b, sigma, tmp, table_slot);
    if (r == TOPO_AMD_OK) r = run_axis1_mfma(tmp, b.out_rows, b.nx, sigma, out, table_slot + 1);
    t_run_if = nullptr;
    t_axis0_one_tile = false;
    return r;
}

int run_axis0(const Block& b, double sigma, float* out, int table_slot, bool mfma_ok = true, bool small_ok = true) {
    Context& c = ctx();
    if (mfma_ok && mfma_radius(gaussian_radius(sigma), b.nx, false, small_ok) && mfma_rows_ok(b, gaussian_radius(sigma)))
        return run_axis0_mfma(b, sigma, out, table_slot);
    const bool wide = wide_tiling(gaussian_radius(sigma));
    GaussArgs a{};
    TOPO_TRY(upload_weights(table_slot, sigma, wide ? 16 : 8, &a));
    a.in = b.in;
    a.out = out;
    a.in_rows = b.in_rows;
    a.in_row0 = b.in_row0;
    a.gny = b.gny;
    a.nx = b.nx;
    a.out_row0 = b.out_row0;
    a.out_rows = b.out_rows;
    const int tb = wide ? 16 : 8;
    a.group0 = b.out_row0 / tb;
    const int groups = (b.out_row0 + b.out_rows - 1) / tb - a.group0 + 1;
    dim3 grid((b.nx + kThreads - 1) / kThreads, groups);
    if (wide) hipLaunchKernelGGL((gauss_axis0_kernel<16, 16>), grid, dim3(kThreads), 0, c.compute, a);
    else hipLaunchKernelGGL((gauss_axis0_kernel<8, 8>), grid, dim3(kThreads), 0, c.compute, a);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

template <int TB, int KB, int NW>
int launch_axis1(const GaussArgs& a, int rows, int nx, double sigma) {
    Context& c = ctx();
    constexpr int tc = NW * TB;
    const int cols_l = tc + a.nchunks * KB - 1;
    const size_t lds_in = (size_t)64 * (cols_l | 1) * sizeof(float);
    const size_t lds_out = (size_t)64 * (tc + 1) * sizeof(float);
    const size_t lds = lds_in > lds_out ? lds_in : lds_out;
    if (lds > 160 * 1024) {
        set_error("gaussian: sigma %.3f (radius %d) needs %zu B of LDS per tile; the "
                  "large-sigma path is not built yet", sigma, a.radius, lds);
        return TOPO_AMD_EUNSUP;
    }
    TOPO_HIP(hipFuncSetAttribute((const void*)gauss_axis1_kernel<TB, KB, NW>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((nx + tc - 1) / tc, (rows + 63) / 64);
    hipLaunchKernelGGL((gauss_axis1_kernel<TB, KB, NW>), grid, dim3(NW * 64), lds, c.compute, a);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

int run_axis1_wave_grad(const float* in, int s_row0, int s_rows, double sigma, const GradArgs& g,
                        float* smooth_out);

// `in` holds exactly the rows [out_row0, out_row0 + out_rows) starting at in_row0 == out_row0
int run_axis1(const float* in, int rows, int nx, double sigma, float* out, int table_slot, bool small_ok = true) {
    if (mfma_radius(gaussian_radius(sigma), nx, false, small_ok)) return run_axis1_mfma(in, rows, nx, sigma, out, table_slot);
    const bool wide = wide_tiling(gaussian_radius(sigma));
    GaussArgs a{};
    TOPO_TRY(upload_weights(table_slot, sigma, wide ? 16 : 8, &a));
    a.in = in;
    a.out = out;
    a.in_rows = rows;
    a.in_row0 = 0;
    a.gny = rows;
    a.nx = nx;
    a.out_row0 = 0;
    a.out_rows = rows;
    a.group0 = 0;
    if (wide) {
        const int cols_l = 8 * 16 + a.nchunks * 16 - 1;
        if ((size_t)64 * (cols_l | 1) * sizeof(float) > 160 * 1024) {
            // long filter: the wave-shift kernel has no LDS tile
            GradArgs g{};
            g.gny = rows;
            g.nx = nx;
            g.out_row0 = 0;
            g.out_rows = rows;
            const int r = run_axis1_wave_grad(in, 0, rows, sigma, g, out);
            if (r != TOPO_AMD_EUNSUP) return r;
            // filter wider than a wavefront can chain: transpose, filter down the columns, transpose back
            Context& c = ctx();
            const size_t bytes = (size_t)rows * nx * sizeof(float);
            void *t1 = nullptr, *t2 = nullptr;
            TOPO_TRY(workspace(6, bytes, &t1));
            TOPO_TRY(workspace(7, bytes, &t2));
            dim3 g1((nx + 31) / 32, (rows + 31) / 32), g2((rows + 31) / 32, (nx + 31) / 32);
            hipLaunchKernelGGL(transpose_kernel, g1, dim3(kThreads), 0, c.compute, in, (float*)t1, rows, nx);
            TOPO_HIP(hipGetLastError());
            Block tb{(const float*)t1, nx, 0, nx, rows, 0, nx};
            TOPO_TRY(run_axis0(tb, sigma, (float*)t2, table_slot));
            hipLaunchKernelGGL(transpose_kernel, g2, dim3(kThreads), 0, c.compute, (const float*)t2, out, nx, rows);
            TOPO_HIP(hipGetLastError());
            return TOPO_AMD_OK;
        }
        return launch_axis1<16, 16, 8>(a, rows, nx, sigma);
    }
    return launch_axis1<8, 8, 8>(a, rows, nx, sigma);
}

template <int TB, int KB, int NW, int PF>
int launch_axis1_grad_pf(const GaussArgs& a, const GradArgs& g, size_t lds);

template <int TB, int KB, int NW>
int launch_axis1_grad(const GaussArgs& a, const GradArgs& g, double sigma) {
    constexpr int tc = NW * TB;
    const int cols_l = tc + a.nchunks * KB - 1;
    const size_t lds_in = (size_t)64 * (cols_l | 1) * sizeof(float);
    const size_t lds_out = (size_t)64 * (tc + 1) * sizeof(float);
    const size_t lds = lds_in > lds_out ? lds_in : lds_out;
    if (lds > 160 * 1024 || cols_l > 320) {
        set_error("gradient: sigma %.3f (radius %d) needs %zu B of LDS and %d columns per tile; "
                  "outside what the LDS-tiled axis-1 kernel is built for", sigma, a.radius, lds, cols_l);
        return TOPO_AMD_EUNSUP;
    }
    if (cols_l > 256) return launch_axis1_grad_pf<TB, KB, NW, 5>(a, g, lds);
    if (cols_l > 192) return launch_axis1_grad_pf<TB, KB, NW, 4>(a, g, lds);
    return launch_axis1_grad_pf<TB, KB, NW, 3>(a, g, lds);
}

template <int TB, int KB, int NW, int PF>
int launch_axis1_grad_pf(const GaussArgs& a, const GradArgs& g, size_t lds) {
    Context& c = ctx();
    constexpr int tc = NW * TB;
    TOPO_HIP(hipFuncSetAttribute((const void*)gauss_axis1_grad_kernel<TB, KB, NW, PF>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    TOPO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)gauss_axis1_grad_kernel<TB, KB, NW, PF>,
                                                          NW * 64, lds));
    if (per_cu < 1) per_cu = 1;
    constexpr int out_r = 62, out_c = tc - 2;
    const int tiles_y = (g.out_row0 + g.out_rows - 1) / out_r - g.out_row0 / out_r + 1;
    const int tiles_x = (g.nx + out_c - 1) / out_c;
    const long ntiles = (long)tiles_x * tiles_y;
    long grid = (long)(c.num_cu - c.reserve_cus) * per_cu;  // persistent: every block walks the tile list
    if (grid > ntiles) grid = ntiles;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((gauss_axis1_grad_kernel<TB, KB, NW, PF>), dim3((unsigned)grid), dim3(NW * 64), lds, c.compute, a, g,
                       tiles_x, (int)ntiles);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

// axis-1 smoothing of the plane `in` (rows [s_row0, s_row0 + s_rows) of the DEM, already
// smoothed along axis 0) fused with the gradient epilogue described by g.
int run_axis1_grad(const float* in, int s_row0, int s_rows, int gny, int nx, double sigma,
                   const GradArgs& g, int table_slot) {
    constexpr int wide_from = 60;
    // 16-wide tap chunks pay from radius ~60: 4.13 (narrow) / 3.92 ms (wide) at radius 64, 3.33 / 3.65
    // at radius 56 on 16384^2 (tools/gauss_fused_sweep.sh)
    const bool wide = gaussian_radius(sigma) >= wide_from;
    GaussArgs a{};
    TOPO_TRY(upload_weights(table_slot, sigma, wide ? 16 : 8, &a));
    a.in = in;
    a.out = nullptr;
    a.in_rows = s_rows;
    a.in_row0 = s_row0;
    a.gny = gny;
    a.nx = nx;
    a.out_row0 = 0;
    a.out_rows = 0;
    a.group0 = 0;
    if (wide) return launch_axis1_grad<16, 16, 8>(a, g, sigma);
    return launch_axis1_grad<16, 8, 8>(a, g, sigma);
}

// Wave-shift axis 1 + epilogue.  TOPO_AMD_EUNSUP when the filter spans too many lanes.
int run_axis1_wave_grad(const float* in, int s_row0, int s_rows, double sigma, const GradArgs& g,
                        float* smooth_out) {
    Context& c = ctx();
    const int R = gaussian_radius(sigma);
    auto fdiv = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
    const int d_lo = -fdiv(R + GC - 1, GC), d_hi = fdiv(R + GC - 1, GC);
    const int nsteps = d_hi - d_lo + 1;
    const int nvl = 64 - (d_hi - d_lo);  // lanes of a wavefront that produce output
    constexpr int min_lanes = 42;
    // below ~42 of 64 output lanes (radius > 176, sigma > ~44) the transpose path is faster:
    // 11.2 vs 9.0 ms at sigma 50, 68 vs 17 ms at sigma 107 on 16384^2 (tools/gauss_long_crossover.py)
    if (nvl < min_lanes) return TOPO_AMD_EUNSUP;
    std::vector<double> w(2 * R + 1);
    double sum = 0.0;
    for (int k = -R; k <= R; ++k) {
        w[k + R] = std::exp(-0.5 / (sigma * sigma) * (double)k * (double)k);
        sum += w[k + R];
    }
    // chain step `step` handles lane offset D = d_hi - step: output sub-column t of lane L takes
    // sample s of lane L + D with tap index GC*D + s - t + R
    std::vector<float> wtab((size_t)nsteps * 32, 0.0f), wsum((size_t)nsteps * GC, 0.0f);
    for (int step = 0; step < nsteps; ++step) {
        const int D = d_hi - step;
        for (int j = 0; j <= 30; ++j) {
            const int k = GC * D + (j - 15) + R;
            if (k >= 0 && k <= 2 * R) wtab[(size_t)step * 32 + j] = (float)(w[k] / sum);
        }
    }
    // wsum[D][t] = taps lane L+D contributes to sub-column t in total; the chain consumes them
    // cumulated: sum_D wsum[D] (v0[L+D] - v0[L]) = sum_{j>=1} eps[L+j] T+[j] - sum_{j<=0} eps[L+j] T-[j],
    // eps[m] = v0[m] - v0[m-1], T+[j] = sum_{D>=j} wsum[D], T-[j] = sum_{D<=j-1} wsum[D]
    for (int t = 0; t < GC; ++t) {
        std::vector<double> ws(nsteps);
        for (int step = 0; step < nsteps; ++step) {
            double acc = 0.0;
            for (int sidx = 0; sidx < GC; ++sidx) acc += (double)wtab[(size_t)step * 32 + sidx - t + 15];
            ws[step] = acc;
        }
        for (int step = 0; step < nsteps; ++step) {
            const int j = d_hi - step;
            double acc = 0.0;
            if (j >= 1) {
                for (int D = j; D <= d_hi; ++D) acc += ws[d_hi - D];
            } else {
                for (int D = d_lo; D <= j - 1; ++D) acc -= ws[d_hi - D];
            }
            wsum[(size_t)step * GC + t] = (float)acc;
        }
    }
    void *d_w = nullptr, *d_s = nullptr;
    TOPO_TRY(upload_table(2, wtab.data(), wtab.size() * sizeof(float), &d_w));
    TOPO_TRY(upload_table(3, wsum.data(), wsum.size() * sizeof(float), &d_s));
    WaveGradArgs a;
    a.in = in;
    a.in_row0 = s_row0;
    a.in_rows = s_rows;
    a.wtab = (const float*)d_w;
    a.wsum = (const float*)d_s;
    a.radius = R;
    a.d_lo = d_lo;
    a.nsteps = nsteps;
    a.nvl = nvl;
    a.out_c = GC * (nvl - 2);
    a.rows_per_wave = 32;
    a.smooth_out = smooth_out;
    a.g = g;
    const int tiles = (g.out_row0 + g.out_rows - 1) / a.rows_per_wave - g.out_row0 / a.rows_per_wave + 1;
    dim3 grid((g.nx + a.out_c - 1) / a.out_c, (tiles + kThreads / 64 - 1) / (kThreads / 64));
    hipLaunchKernelGGL(gauss_axis1_wave_grad_kernel, grid, dim3(kThreads), 0, c.compute, a);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

// Full 2-D smooth of rows [row0, row0+rows) into `out`; ws_slot names the scratch plane.
int smooth_rows(const Block& src, double sigma_y, double sigma_x, int row0, int rows, float* out,
                int ws_slot, int table_slot, bool small_ok = true) {
    Context& c = ctx();
    const bool do_y = sigma_y > 1e-15, do_x = sigma_x > 1e-15;
    Block b = src;
    b.out_row0 = row0;
    b.out_rows = rows;
    const size_t bytes = (size_t)rows * src.nx * sizeof(float);
    if (do_y && do_x) {
        void* tmp = nullptr;
        TOPO_TRY(workspace(ws_slot, bytes, &tmp));
        const int R = gaussian_radius(sigma_y);
        if (sigma_y == sigma_x && mfma_radius(R, b.nx, false, small_ok) && mfma_rows_ok(b, R)) {
            TOPO_TRY(smooth_both_mfma(b, sigma_y, (float*)tmp, out, table_slot, false));
        } else {
            TOPO_TRY(run_axis0(b, sigma_y, (float*)tmp, table_slot, true, small_ok));
            TOPO_TRY(run_axis1((const float*)tmp, rows, src.nx, sigma_x, out, table_slot + 1, small_ok));
        }
    } else if (do_y) {
        TOPO_TRY(run_axis0(b, sigma_y, out, table_slot, true, small_ok));
    } else if (do_x) {
        const float* first = src.in + (size_t)(row0 - src.in_row0) * src.nx;
        TOPO_TRY(run_axis1(first, rows, src.nx, sigma_x, out, table_slot + 1, small_ok));
    } else {
        const float* first = src.in + (size_t)(row0 - src.in_row0) * src.nx;
        TOPO_HIP(hipMemcpyAsync(out, first, bytes, hipMemcpyDeviceToDevice, c.compute));
    }
    return TOPO_AMD_OK;
}

int upload_resolution(int res_mode, const void* res_x, const void* res_y, int nx, int gny,
                      const float** dx, const float** dy) {
    if (res_mode == TOPO_AMD_RES_2D) {
        *dx = (const float*)res_x;
        *dy = (const float*)res_y;
        return TOPO_AMD_OK;
    }
    const int n_x = res_mode == TOPO_AMD_RES_SCALAR ? 1 : nx;
    const int n_y = res_mode == TOPO_AMD_RES_SCALAR ? 1 : gny;
    std::vector<float> fx(n_x), fy(n_y);
    const double* hx = (const double*)res_x;
    const double* hy = (const double*)res_y;
    for (int i = 0; i < n_x; ++i) fx[i] = (float)hx[i];
    for (int i = 0; i < n_y; ++i) fy[i] = (float)hy[i];
    void *d0 = nullptr, *d1 = nullptr;
    TOPO_TRY(upload_table(4, fx.data(), fx.size() * sizeof(float), &d0));
    TOPO_TRY(upload_table(5, fy.data(), fy.size() * sizeof(float), &d1));
    *dx = (const float*)d0;
    *dy = (const float*)d1;
    return TOPO_AMD_OK;
}

}  // namespace

int gaussian_radius(double sigma) { return (int)(4.0 * sigma + 0.5); }

int mfma_min_radius(bool for_gradient) { return mfma_min_radius_impl(for_gradient); }

// (for the call: the matrix-core routes are closed to a raster whose samples are mostly beyond their range - a property of
// the WHOLE raster, common.hpp RasterClass, so that every row block of it takes the same kernels)
struct NoMfmaScope {
    bool before;
    explicit NoMfmaScope(const Block&) : before(t_no_mfma) {
        static const bool on = [] {  // TOPO_AMD_GAUSS_LARGE_SAMPLE=0: never (A/B: tools/large_raster_time.py)
            const char* e = std::getenv("TOPO_AMD_GAUSS_LARGE_SAMPLE");
            return !(e && *e == '0');
        }();
        if (on && !t_no_mfma) t_no_mfma = current_class().large;
    }
    ~NoMfmaScope() { t_no_mfma = before; }
};
int launch_gaussian(const Block& b, double sigma_y, double sigma_x, float* out, bool small_ok) {
    NoMfmaScope scope(b);
    for (int r = 0; r < b.out_rows; r += kMaxLaunchRows) {  // (one pass unless the block is taller than a launch covers)
        const int n = std::min(kMaxLaunchRows, b.out_rows - r);
        TOPO_TRY(smooth_rows(b, sigma_y, sigma_x, b.out_row0 + r, n, out + (size_t)r * b.nx, 0, 1, small_ok));
    }
    return TOPO_AMD_OK;
}

int launch_sobel(const Block& b, float* dx_out, float* dy_out) {
    if (b.out_rows > kMaxLaunchRows) {
        for (int r = 0; r < b.out_rows; r += kMaxLaunchRows) {
            Block s = b;
            s.out_row0 = b.out_row0 + r;
            s.out_rows = std::min(kMaxLaunchRows, b.out_rows - r);
            TOPO_TRY(launch_sobel(s, dx_out ? dx_out + (size_t)r * b.nx : nullptr, dy_out ? dy_out + (size_t)r * b.nx : nullptr));
        }
        return TOPO_AMD_OK;
    }
    Context& c = ctx();
    GradArgs g{};
    g.raw = b.in;
    g.raw_rows = b.in_rows;
    g.raw_row0 = b.in_row0;
    g.gny = b.gny;
    g.nx = b.nx;
    g.out_row0 = b.out_row0;
    g.out_rows = b.out_rows;
    g.dx = dx_out;
    g.dy = dy_out;
    TOPO_TRY(check_grid_rows(b.out_rows, "sobel / gradient epilogue"));
    dim3 grid((b.nx + kThreads - 1) / kThreads, b.out_rows);
    hipLaunchKernelGGL(sobel_kernel<false>, grid, dim3(kThreads), 0, c.compute, g);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

namespace {
// the rows the gradient smooths: the output rows plus one neighbour row inside the DEM
Block smoothed_rows_block(const Block& b) {
    Block r = b;
    r.out_row0 = b.out_row0 > 0 ? b.out_row0 - 1 : 0;
    const int s1 = (b.out_row0 + b.out_rows + 1 < b.gny) ? b.out_row0 + b.out_rows + 1 : b.gny;
    r.out_rows = s1 - r.out_row0;
    return r;
}
}  // namespace

int launch_gradient(const Block& b, double sigma, double sig_ratio, int res_mode,
                    const void* res_x, const void* res_y, float* dx, float* dy, float* slope,
                    float* aspect) {
    NoMfmaScope scope(b);
    if (b.out_rows > kMaxLaunchRows) {
        if (ctx().ghost.armed) return TOPO_AMD_EUNSUP;
        for (int r = 0; r < b.out_rows; r += kMaxLaunchRows) {
            Block s = b;
            s.out_row0 = b.out_row0 + r;
            s.out_rows = std::min(kMaxLaunchRows, b.out_rows - r);
            const size_t shift = (size_t)r * b.nx;
            const void *rx = res_x, *ry = res_y;
            if (res_mode == TOPO_AMD_RES_2D && res_x && res_y) {
                rx = (const float*)res_x + shift;
                ry = (const float*)res_y + shift;
            }
            TOPO_TRY(launch_gradient(s, sigma, sig_ratio, res_mode, rx, ry, dx ? dx + shift : nullptr, dy ? dy + shift : nullptr,
                                     slope ? slope + shift : nullptr, aspect ? aspect + shift : nullptr));
        }
        return TOPO_AMD_OK;
    }
    Context& c = ctx();
    TOPO_REQUIRE(res_mode >= 0 && res_mode <= 2, "gradient: bad res_mode %d", res_mode);
    TOPO_REQUIRE(res_x && res_y, "gradient: resolution arrays are NULL");
    // a row shard whose exchange is in flight (common.hpp, GhostGate): only the chunked matrix-core route below knows
    // what to do with it; every other route says so before it launches anything
    static const int chunk_min = [] {
        const char* e = std::getenv("TOPO_AMD_GRAD_CHUNK_MIN_ROWS");
        return e && *e ? std::atoi(e) : 2048;  // (the interior of a 4096-row shard of the 8-GPU split goes in two chunks)
    }();
    const bool gated = c.ghost.armed;
    if (gated && !(sigma > 1.0 && sig_ratio == 1.0 && mfma_radius(gaussian_radius(sigma), b.nx, true) &&
                   mfma_rows_ok(smoothed_rows_block(b), gaussian_radius(sigma)) && b.out_rows >= chunk_min))
        return TOPO_AMD_EUNSUP;
    GradArgs g{};
    TOPO_TRY(upload_resolution(res_mode, res_x, res_y, b.nx, b.gny, &g.res_x, &g.res_y));
    g.res_mode = res_mode;
    g.gny = b.gny;
    g.nx = b.nx;
    g.out_row0 = b.out_row0;
    g.out_rows = b.out_rows;
    g.dx = dx;
    g.dy = dy;
    g.slope = slope;
    g.aspect = aspect;
    TOPO_TRY(check_grid_rows(b.out_rows, "sobel / gradient epilogue"));
    dim3 grid((b.nx + kThreads - 1) / kThreads, b.out_rows);
    if (sigma <= 1.0) {  // topo.py:628-629
        g.raw = b.in;
        g.raw_rows = b.in_rows;
        g.raw_row0 = b.in_row0;
        hipLaunchKernelGGL(sobel_kernel<true>, grid, dim3(kThreads), 0, c.compute, g);
        TOPO_HIP(hipGetLastError());
        return TOPO_AMD_OK;
    }
    TOPO_REQUIRE(b.gny >= 2 && b.nx >= 2,
                 "gradient: numpy.gradient needs at least 2 samples per axis (got %d x %d)",
                 b.gny, b.nx);
    // Matrix-core route on a large block: the smooth is MFMA-bound and leaves HBM idle, the epilogue is
    // HBM-bound and needs no LDS, so the rows go in chunks and the epilogue of chunk k runs on a second stream
    // next to the smooth of chunk k + 1 (15.7 -> 14.5 ms at sigma 30.25 on 32768^2 with 4 -> 8 chunks; one chunk:
    // 18.5).  Row chunks are row blocks:
    // same bits.
    if (sigma > 1.0 && sig_ratio == 1.0 && mfma_radius(gaussian_radius(sigma), b.nx, true) &&
        mfma_rows_ok(smoothed_rows_block(b), gaussian_radius(sigma)) && b.out_rows >= chunk_min) {
        // (widths that are not multiples of 4, or planes that are not 16-byte aligned: the one-pixel-per-thread epilogue)
        const bool wide_epilogue = b.nx >= 8;
        static const int NCH = [] {
            const char* e = std::getenv("TOPO_AMD_GRAD_CHUNKS");
            return std::max(1, std::min(64, e && *e ? std::atoi(e) : 8));
        }();
        static const int chunk_rows = [] {  // smallest chunk (rows); a chunk's planes should fit the 256 MB Infinity Cache
            const char* e = std::getenv("TOPO_AMD_GRAD_CHUNK_ROWS");
            // 32768^2, chunks x rows (profiles/r03_gradient_chunks.txt): 8 x 4096 8.04 / 13.70 ms at sigma 3.25 / 30.25,
            // 16 x 2048 7.77 / 13.50, 32 x 1024 8.27 / 14.55, 64 x 512 8.54 / 16.16
            // with the f16 kernels (profiles/r03_gauss_f16.txt, section 7): 4 x 8192 6.15 / 9.01, 8 x 4096 6.08 / 9.12,
            // 12 x 2752 6.31 / 9.40, 16 x 2048 6.31 / 9.59, 32 x 1024 6.64 / 11.20; one stream: 6.71 / 10.43
            return std::max(256, e && *e ? std::atoi(e) : 4096);
        }();
        constexpr bool use_aux = true;
        // A short block (a row shard: 4096 rows) in few chunks leaves the smooth of its first chunk and the epilogue of its
        // last one uncovered: the fused route (one kernel per chunk, nothing restaged across a cut but Rp rows) goes in at
        // least 6 chunks of 512 rows or more, the two-pass route (every chunk restages 2 R rows on axis 0: 24 % at
        // radius 121 and 1024 rows) in at least 3.  One 4096-row shard in loop-back, sigma 3.25 / 30.25, ms per step
        // (profiles/r04_shard_fused.txt): 2 chunks 0.98 / 1.56, 3: 0.95 / 1.55, 4: 0.83 / 1.60, 6: 0.81 / 1.73, 8: 0.85 / 1.91.
        constexpr int kMinChunksEnv = 0;
        const int kMinChunks = kMinChunksEnv ? kMinChunksEnv
                                             : std::max(2, std::min(fused_radius(gaussian_radius(sigma), true) ? 6 : 3, b.out_rows / 512));
        if (!c.aux) {
            TOPO_HIP(hipStreamCreateWithFlags(&c.aux, hipStreamNonBlocking));
            for (auto& e : c.aux_ready) TOPO_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            TOPO_HIP(hipEventCreateWithFlags(&c.aux_done, hipEventDisableTiming));
        }
        // smoothed rows [s0, s1) = the output rows plus one neighbour row inside the DEM, cut at global multiples
        // of 32 rows (the row tiles of the axis-0 kernel; a chunk of 32 k rows is also a whole number of axis-1
        // bands).  Behind the smooth of a chunk, the epilogue emits the output rows whose two neighbour rows are smoothed
        // by then: the chunk's own rows, short of the first / last one where the chunk next to it is still to come,
        // plus the row that chunk left behind where it is done.
        const int s0 = b.out_row0 > 0 ? b.out_row0 - 1 : 0;
        const int s1 = (b.out_row0 + b.out_rows + 1 < b.gny) ? b.out_row0 + b.out_rows + 1 : b.gny;
        const int out_end = b.out_row0 + b.out_rows;
        const size_t bytes = (size_t)(s1 - s0) * b.nx * sizeof(float);
        void *pa = nullptr, *pb = nullptr;
        TOPO_TRY(workspace(1, bytes, &pa));
        TOPO_TRY(workspace(2, bytes, &pb));
        // chunks of >= 2048 rows (every chunk restages the 2 R halo rows of the axis-0 ring: 12 % at radius 121); a
        // short block (a row shard) goes in at least kMinChunks, because the epilogue of its LAST chunk runs alone
        const int nch = std::max(kMinChunks, std::min(NCH, (s1 - s0) / chunk_rows));
        const int per = std::max(32, ((s1 - s0 + nch - 1) / nch + 31) / 32 * 32);
        struct Chunk {
            int c0, c1;
            bool ghost, done;
        };
        std::vector<Chunk> chunks;
        const int R = gaussian_radius(sigma);
        // Taper (TOPO_AMD_GRAD_TAPER, default on from 4 chunks): the smooth of the chunk processed FIRST and the epilogue of
        // the one processed LAST run uncovered, so those two chunks are a quarter of the others (>= 256 rows).  Processing
        // order = row order, except in a row shard, where the first chunk processed is the second in row order (the top
        // one waits for the exchange).
        constexpr bool taper_on = true;
        // (32768^2, sigma 3.25: 5.76 -> 5.52 ms; a 4096-row shard in 6 chunks LOSES 7 % with a 256-row first chunk, so only
        // blocks whose chunks are 2048 rows and more are tapered; sigma 30.25: no difference either way)
        const bool taper = taper_on && nch >= 4 && (s1 - s0) / nch >= 2048;
        const int small = std::max(256, ((s1 - s0) / nch / 4 + 31) / 32 * 32);
        const int big = taper ? std::max(32, ((s1 - s0 - 2 * small + nch - 3) / (nch - 2) + 31) / 32 * 32) : per;
        const int first_small = gated ? 1 : 0;  // row-order index of the chunk processed first
        for (int c0 = s0, k = 0; c0 < s1; ++k) {
            const int want = taper && k == first_small ? small : big;
            int c1 = (c0 + want) / 32 * 32;
            if (taper && s1 - c1 < small + 32 && s1 - c1 > 0 && c1 < s1) c1 = std::max(c0 + 32, (s1 - small) / 32 * 32);  // leave the small last chunk
            if (c1 + 32 > s1 || c1 <= c0) c1 = s1;  // no sliver at the end
            // row shard with its exchange in flight: does the filter of these rows reach into the ghost rows?
            chunks.push_back({c0, c1, gated && (c0 - R < c.ghost.ghost_lo || c1 + R > c.ghost.ghost_hi), false});
            c0 = c1;
        }
        TOPO_REQUIRE(chunks.size() <= 64, "gradient: %zu row chunks (at most 64)", chunks.size());
        // Row shard: the chunks that stay clear of the ghost rows go first, on the rows the shard owns (the ghost rows are
        // being written meanwhile; staging clamps to the view), then the compute stream waits for the exchange and the
        // chunks at the two seams follow on the whole block - ordinary chunks of the ordinary pipeline, no seam strips
        // of their own, and the exchange hides behind the first chunks.  Row chunks are row blocks: same bits.
        std::vector<int> order;
        for (int pass = 0; pass < 2; ++pass)
            for (int k = 0; k < (int)chunks.size(); ++k)
                if (chunks[k].ghost == (pass == 1)) order.push_back(k);
        Block owned = b;
        if (gated) {
            const int lo = std::max(b.in_row0, c.ghost.ghost_lo), hi = std::min(b.in_row0 + b.in_rows, c.ghost.ghost_hi);
            owned.in = b.in + (size_t)(lo - b.in_row0) * b.nx;
            owned.in_row0 = lo;
            owned.in_rows = hi - lo;
            c.ghost.armed = false;  // taken
        }
        // one "met a sample that is not a plain finite one" flag for the call: the fused kernels of all chunks raise it, and
        // the two-pass route (five launches that return at once on an ordinary DEM) is queued ONCE, behind the last
        // chunk, over all rows - not behind every chunk, where its launches sat in front of the next chunk's smooth
        void* wild = nullptr;
        TOPO_TRY(workspace(11, 64, &wild));
        TOPO_HIP(hipMemsetAsync(wild, 0, sizeof(int), c.compute));
        bool owed = false;
        bool waited = !gated;
        for (int k : order) {
            Chunk& ck = chunks[k];
            if (ck.ghost && !waited) {
                TOPO_TRY(topo_amd_halo_wait());
                waited = true;
            }
            Block rows = ck.ghost || !gated ? b : owned;
            rows.out_row0 = ck.c0;
            rows.out_rows = ck.c1 - ck.c0;
            float* a_k = (float*)pa + (size_t)(ck.c0 - s0) * b.nx;
            float* b_k = (float*)pb + (size_t)(ck.c0 - s0) * b.nx;
            bool owes = false;
            TOPO_TRY(smooth_both_mfma(rows, sigma, a_k, b_k, 1, true, (const int*)wild, &owes));
            owed = owed || owes;
            ck.done = true;
            if (use_aux) {
                TOPO_HIP(hipEventRecord(c.aux_ready[k], c.compute));
                TOPO_HIP(hipStreamWaitEvent(c.aux, c.aux_ready[k], 0));
            }
            const bool first = k == 0, last = k + 1 == (int)chunks.size();
            const int o0 = std::max(b.out_row0, first ? b.out_row0 : (chunks[k - 1].done ? ck.c0 - 1 : ck.c0 + 1));
            const int o1 = std::min(out_end, last ? out_end : (chunks[k + 1].done ? ck.c1 + 1 : ck.c1 - 1));
            if (o1 > o0) {
                GradArgs gk = g;
                const size_t shift = (size_t)(o0 - b.out_row0) * b.nx;
                gk.out_row0 = o0;
                gk.out_rows = o1 - o0;
                gk.dx = dx ? dx + shift : nullptr;
                gk.dy = dy ? dy + shift : nullptr;
                gk.slope = slope ? slope + shift : nullptr;
                gk.aspect = aspect ? aspect + shift : nullptr;
                if (res_mode == TOPO_AMD_RES_2D) {
                    gk.res_x = g.res_x + shift;
                    gk.res_y = g.res_y + shift;
                }
                gk.gx_src = (const float*)pb;
                gk.gy_src = (const float*)pb;
                gk.s_row0 = s0;
                gk.s_rows = s1 - s0;
                TOPO_TRY(check_grid_rows(o1 - o0, "gradient epilogue"));
                if (wide_epilogue) {
                    dim3 grid4(((b.nx + 3) / 4 + kThreads - 1) / kThreads, o1 - o0);
                    hipLaunchKernelGGL(gradient_epilogue4_kernel, grid4, dim3(kThreads), 0, use_aux ? c.aux : c.compute, gk);
                } else {
                    dim3 grid1((b.nx + kThreads - 1) / kThreads, o1 - o0);
                    hipLaunchKernelGGL(gradient_epilogue_kernel, grid1, dim3(kThreads), 0, use_aux ? c.aux : c.compute, gk);
                }
                TOPO_HIP(hipGetLastError());
            }
        }
        if (!waited) TOPO_TRY(topo_amd_halo_wait());
        if (owed) {
            // behind every chunk and its epilogue: smooth [s0, s1) again by the two passes and redo the epilogue, if the flag is up
            if (use_aux) {
                TOPO_HIP(hipEventRecord(c.aux_done, c.aux));
                TOPO_HIP(hipStreamWaitEvent(c.compute, c.aux_done, 0));
            }
            Block rows = b;
            rows.out_row0 = s0;
            rows.out_rows = s1 - s0;
            TOPO_TRY(smooth_two_pass_if(rows, sigma, (float*)pa, (float*)pb, 1, (const int*)wild));
            GradArgs gk = g;
            gk.gx_src = (const float*)pb;
            gk.gy_src = (const float*)pb;
            gk.s_row0 = s0;
            gk.s_rows = s1 - s0;
            gk.run_if = (const int*)wild;
            const int if_rows = std::min(b.out_rows, 512);
            if (wide_epilogue) {
                dim3 grid4(((b.nx + 3) / 4 + kThreads - 1) / kThreads, if_rows);
                hipLaunchKernelGGL(gradient_epilogue4_if_kernel, grid4, dim3(kThreads), 0, c.compute, gk);
            } else {
                dim3 grid1((b.nx + kThreads - 1) / kThreads, if_rows);
                hipLaunchKernelGGL(gradient_epilogue_if_kernel, grid1, dim3(kThreads), 0, c.compute, gk);
            }
            TOPO_HIP(hipGetLastError());
            return TOPO_AMD_OK;
        }
        if (use_aux) {
            TOPO_HIP(hipEventRecord(c.aux_done, c.aux));
            TOPO_HIP(hipStreamWaitEvent(c.compute, c.aux_done, 0));
        }
        return TOPO_AMD_OK;
    }
    // smoothed rows needed: the output rows plus one neighbour row inside the DEM
    const int s0 = b.out_row0 > 0 ? b.out_row0 - 1 : 0;
    const int s1 = (b.out_row0 + b.out_rows + 1 < b.gny) ? b.out_row0 + b.out_rows + 1 : b.gny;
    const int s_rows = s1 - s0;
    const size_t bytes = (size_t)s_rows * b.nx * sizeof(float);
    void *plane_a = nullptr, *plane_b = nullptr;
    TOPO_TRY(workspace(1, bytes, &plane_a));
    if (sig_ratio == 1.0) {  // topo.py:630-631: one smooth; axis 1 and the epilogue run fused
        Block rows = b;
        rows.out_row0 = s0;
        rows.out_rows = s_rows;
        const bool mfma = mfma_radius(gaussian_radius(sigma), b.nx, true) && mfma_rows_ok(rows, gaussian_radius(sigma));
        if (!mfma) TOPO_TRY(run_axis0(rows, sigma, (float*)plane_a, 1, false));
        // short and medium filters: LDS-tiled axis 1 (9.8 vs 13.6 ms at sigma 3.25 on 32768^2); long
        // filters: wave-shift axis 1 (26.7 vs 27.8 ms at sigma 30.25) while enough lanes produce output
        constexpr int fused_max = 92;
        // the LDS-tiled fused kernel while its tile fits (320 columns: radius 92), the wave-shift one
        // beyond: 2.67 vs 3.63 ms at radius 32, 3.33 vs 4.07 at 56, 4.60 vs 5.17 at 92 on 16384^2
        if (!mfma) {
            if (gaussian_radius(sigma) <= fused_max) {
                const int r = run_axis1_grad((const float*)plane_a, s0, s_rows, b.gny, b.nx, sigma, g, 2);
                if (r != TOPO_AMD_EUNSUP) return r;
            }
            const int r = run_axis1_wave_grad((const float*)plane_a, s0, s_rows, sigma, g, nullptr);
            if (r != TOPO_AMD_EUNSUP) return r;
        }
        // matrix-core range: the smooth of the chunked route above (same kernels, same bits), then the stand-alone
        // epilogue; a filter wider than a wavefront can chain: finish the smooth unfused (beyond radius 121 axis 1 goes
        // through the wave-shift or the transpose path)
        TOPO_TRY(workspace(2, bytes, &plane_b));
        if (mfma) TOPO_TRY(smooth_both_mfma(rows, sigma, (float*)plane_a, (float*)plane_b, 1, true));
        else TOPO_TRY(run_axis1((const float*)plane_a, s_rows, b.nx, sigma, (float*)plane_b, 2));
        plane_a = plane_b;
    } else {  // topo.py:633-635
        const double perp = sigma * sig_ratio;
        TOPO_TRY(workspace(2, bytes, &plane_b));
        // (two radii, one ghost depth: the matrix-core kernels only from radius 16, where they need no more
        // ghost rows than the filter itself)
        TOPO_TRY(smooth_rows(b, perp, sigma, s0, s_rows, (float*)plane_a, 0, 1, /*small_ok=*/false));
        TOPO_TRY(smooth_rows(b, sigma, perp, s0, s_rows, (float*)plane_b, 0, 1, /*small_ok=*/false));
    }
    g.gx_src = (const float*)plane_a;
    g.gy_src = (const float*)plane_b;
    g.s_row0 = s0;
    g.s_rows = s_rows;
    if (b.nx >= 8) {
        dim3 grid4(((b.nx + 3) / 4 + kThreads - 1) / kThreads, b.out_rows);
        hipLaunchKernelGGL(gradient_epilogue4_kernel, grid4, dim3(kThreads), 0, c.compute, g);
    } else {
        hipLaunchKernelGGL(gradient_epilogue_kernel, grid, dim3(kThreads), 0, c.compute, g);
    }
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

}  // namespace topo
