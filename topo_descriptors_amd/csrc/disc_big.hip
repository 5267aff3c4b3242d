// K1/K2 for discs too large for an LDS tile (more than ~120 px across): column prefix sums kept
// in HBM as float64, one prefix difference per column offset per pixel.
//
// The reference handles any size with the same FFT (topo.py:175); its example script goes up
// to 100 km scales, i.e. discs of ~2000 px (scripts/compute_topo_descriptors.py:25-38).  This
// path is the completeness fallback: O(size) reads per pixel from L2/HBM instead of O(1) LDS
// traffic, but exact, and it reproduces the zero padding by clamping the prefix rows at the DEM
// edge and skipping columns outside it.  The gather is bound by what it reads (17.6 of 21 ms at
// 151 px on 8192^2 with float64 planes), so the planes are as narrow as exactness allows:
//   * trunc(x): uint32 running prefix that may wrap - a column run of <= 32767 samples of
//     |trunc(x)| <= 65536 fits int32, so the difference of two prefix rows is exact modulo 2^32;
//   * the fractional parts in units of 2^-16 m (the quantisation of the wave-shift kernels), the
//     same way, read only when the block has a fractional sample at all;
//   * trunc(x)^2 for STD: float64 (exact below 2^53; 32 bits are not enough for long runs).
// A block with a non-finite sample or |trunc(x)| > 65536 takes float64 planes for everything: the prefix pass raises a
// flag that the host reads before the gather.  Those planes are exact integers as well - trunc(x) (below 2^24: a sample
// beyond that, or not finite, is MISSING, staged as 0 and counted in a plane of its own), the fractional parts, and
// trunc(x)^2 in two planes (the bits above and below 2^24: either sum stays below 2^53 down any column) - so a prefix
// difference does not depend on where the planes start, i.e. on the row block: a pixel whose disc holds a missing sample
// is NaN (exactly those pixels), every other pixel is the rounding of exact sums - by the expression of the narrow planes
// where sum trunc(x)^2 fits float64 (the same bits as a block without such samples gives), in 128-bit integers beyond.
#include "common.hpp"

namespace topo {

namespace {

constexpr int kThreads = 256;

constexpr float kIntLimit = 65536.0f;  // |trunc(x)| up to which the uint32 planes are exact
enum BigFlags { kBigBad = 1, kBigFrac = 2 };

struct BigArgs {
    const float* in;
    double* qt;   // prefix over rows of trunc(x),   (rows + 1) x nx
    double* qf;   // prefix of the fractional parts in units of 2^-16 m
    double* qt2;  // prefix of trunc(x)^2 (only when STD is wanted); float64 planes: of its bits from 2^24 up
    double* qt2b; // float64 planes: prefix of the low 24 bits of trunc(x)^2
    double* qc;   // float64 planes: prefix of the count of missing samples
    uint32_t* it;   // the same as uint32 running sums (wrap-around allowed): trunc(x)
    uint32_t* ifr;  // fractional parts in units of 2^-16 m
    int* flags;     // BigFlags, raised by big_prefix_int_kernel
    uint32_t *seg_t, *seg_f;  // per segment of kSeg rows and column: its total, then the sum of those above
    double* seg_t2;
    const int* runs;  // per column offset di (size entries): lo | hi << 16, biased by +32768
    int in_row0, gny, nx;
    int p_row0, p_rows;  // rows covered by the prefix planes
    int out_row0, out_rows;
    int off_min, size, centre, taps;
    float *tpi, *sd;
};

// one thread per column walks down the rows: exact running sums
__global__ __launch_bounds__(kThreads) void big_prefix_kernel(BigArgs p) {
    const int x = blockIdx.x * kThreads + threadIdx.x;
    if (x >= p.nx) return;
    double st = 0.0, sf = 0.0, st2 = 0.0, st2b = 0.0, sc = 0.0;
    p.qt[x] = 0.0;
    p.qf[x] = 0.0;
    p.qc[x] = 0.0;
    if (p.qt2) p.qt2[x] = p.qt2b[x] = 0.0;
    const float* src = p.in + (size_t)(p.p_row0 - p.in_row0) * p.nx + x;
    for (int r = 0; r < p.p_rows; ++r) {
        const float v = src[(size_t)r * p.nx];
        const bool missing = !(fabsf(truncf(v)) < 16777216.0f);  // not finite, or beyond +-2^24
        const float t = missing ? 0.0f : truncf(v);
        st += (double)t;
        sf += missing ? 0.0 : (double)rintf((v - t) * 65536.0f);  // the units of the narrow planes: finite sums agree bit for bit
        sc += missing ? 1.0 : 0.0;
        const size_t o = (size_t)(r + 1) * p.nx + x;
        p.qt[o] = st;
        p.qf[o] = sf;
        p.qc[o] = sc;
        if (p.qt2) {
            const double t2 = (double)t * (double)t;  // below 2^48: exact
            const double hi = floor(t2 * (1.0 / 16777216.0));
            st2 += hi;
            st2b += t2 - hi * 16777216.0;
            p.qt2[o] = st2;
            p.qt2b[o] = st2b;
        }
    }
}

// The narrow planes: uint32 running sums of trunc(x) and of the fractional parts, float64 of
// trunc(x)^2.  A column is cut into segments of kSeg rows that are summed in parallel (one thread
// per column and segment: a single thread per column would walk the whole DEM height alone), the
// segment totals are scanned down each column, and a last pass adds them to the rows of the
// segments below the first.  Integer sums do not depend on how they are split.
constexpr int kSeg = 256;

__global__ __launch_bounds__(kThreads) void big_prefix_int_kernel(BigArgs p) {
    const int x = blockIdx.x * kThreads + threadIdx.x;
    const int seg = blockIdx.y;
    if (x >= p.nx) return;
    uint32_t st = 0, sf = 0;
    double st2 = 0.0;
    int flags = 0;
    if (seg == 0) {
        p.it[x] = 0;
        p.ifr[x] = 0;
        if (p.qt2) p.qt2[x] = 0.0;
    }
    const int r0 = seg * kSeg, r1 = min(r0 + kSeg, p.p_rows);
    const float* src = p.in + (size_t)(p.p_row0 - p.in_row0) * p.nx + x;
    for (int r = r0; r < r1; ++r) {
        const float v = src[(size_t)r * p.nx];
        const float t = truncf(v);
        if (!(fabsf(t) <= kIntLimit)) flags |= kBigBad;  // also NaN / inf
        if (v != t) flags |= kBigFrac;
        st += (uint32_t)(int)t;
        sf += (uint32_t)(int)rintf((v - t) * 65536.0f);
        const size_t o = (size_t)(r + 1) * p.nx + x;
        p.it[o] = st;
        p.ifr[o] = sf;
        if (p.qt2) {
            st2 += (double)t * (double)t;
            p.qt2[o] = st2;
        }
    }
    const size_t so = (size_t)seg * p.nx + x;
    p.seg_t[so] = st;
    p.seg_f[so] = sf;
    if (p.qt2) p.seg_t2[so] = st2;
    if (flags) atomicOr(p.flags, flags);
}

// exclusive scan of the segment totals down each column (a few dozen entries)
__global__ __launch_bounds__(kThreads) void big_segment_scan_kernel(BigArgs p, int nseg) {
    const int x = blockIdx.x * kThreads + threadIdx.x;
    if (x >= p.nx) return;
    uint32_t st = 0, sf = 0;
    double st2 = 0.0;
    for (int s = 0; s < nseg; ++s) {
        const size_t so = (size_t)s * p.nx + x;
        const uint32_t a = p.seg_t[so], b = p.seg_f[so];
        p.seg_t[so] = st;
        p.seg_f[so] = sf;
        st += a;
        sf += b;
        if (p.qt2) {
            const double c2 = p.seg_t2[so];
            p.seg_t2[so] = st2;
            st2 += c2;
        }
    }
}

__global__ __launch_bounds__(kThreads) void big_segment_add_kernel(BigArgs p) {
    const int x = blockIdx.x * kThreads + threadIdx.x;
    const int r = kSeg + blockIdx.y;  // rows of the first segment are final already
    if (x >= p.nx) return;
    const size_t so = (size_t)(r / kSeg) * p.nx + x, o = (size_t)(r + 1) * p.nx + x;
    p.it[o] += p.seg_t[so];
    p.ifr[o] += p.seg_f[so];
    if (p.qt2) p.qt2[o] += p.seg_t2[so];
}

// Four adjacent pixels per thread: for a column offset they need four adjacent prefix entries, one
// 16-byte load per plane and prefix row instead of four 4-byte ones (rows are only 4-byte aligned:
// the packed vector types below make the loads unaligned-safe).
constexpr int kPx = 4;
struct __attribute__((packed, aligned(4))) U32x4 { uint32_t v[kPx]; };
struct __attribute__((packed, aligned(8))) F64x4 { double v[kPx]; };

template <bool WANT_TPI, bool WANT_STD, bool FRAC>
__global__ __launch_bounds__(kThreads) void big_disc_int_kernel(BigArgs p) {
    const int ox = (blockIdx.x * kThreads + threadIdx.x) * kPx;
    const int oy = p.out_row0 + blockIdx.y;
    if (ox >= p.nx) return;
    long long st[kPx] = {}, sf[kPx] = {};
    double st2[kPx] = {};
    for (int k = 0; k < p.size; ++k) {
        const unsigned packed = (unsigned)p.runs[k];  // wave-uniform
        const int lo = (int)(packed & 0xffffu) - 32768, hi = (int)(packed >> 16) - 32768;
        const int top = min(max(oy + hi + 1, 0), p.gny) - p.p_row0;
        const int bot = min(max(oy + lo, 0), p.gny) - p.p_row0;
        if (top <= bot) continue;
        const int x = ox + p.off_min + k;
        if (x + kPx <= 0 || x >= p.nx) continue;  // zero padding: columns outside contribute nothing
        const size_t a = (size_t)top * p.nx + x, b = (size_t)bot * p.nx + x;
        if (x >= 0 && x + kPx <= p.nx) {
            const U32x4 ta = *reinterpret_cast<const U32x4*>(p.it + a), tb = *reinterpret_cast<const U32x4*>(p.it + b);
#pragma unroll
            for (int j = 0; j < kPx; ++j) st[j] += (int)(ta.v[j] - tb.v[j]);  // exact: the run sum fits int32
            if (FRAC) {
                const U32x4 fa = *reinterpret_cast<const U32x4*>(p.ifr + a), fb = *reinterpret_cast<const U32x4*>(p.ifr + b);
#pragma unroll
                for (int j = 0; j < kPx; ++j) sf[j] += (int)(fa.v[j] - fb.v[j]);
            }
            if (WANT_STD) {
                const F64x4 qa = *reinterpret_cast<const F64x4*>(p.qt2 + a), qb = *reinterpret_cast<const F64x4*>(p.qt2 + b);
#pragma unroll
                for (int j = 0; j < kPx; ++j) st2[j] += qa.v[j] - qb.v[j];
            }
        } else {  // the DEM's left or right edge cuts the four columns
#pragma unroll
            for (int j = 0; j < kPx; ++j) {
                if (x + j < 0 || x + j >= p.nx) continue;
                st[j] += (int)(p.it[a + j] - p.it[b + j]);
                if (FRAC) sf[j] += (int)(p.ifr[a + j] - p.ifr[b + j]);
                if (WANT_STD) st2[j] += p.qt2[a + j] - p.qt2[b + j];
            }
        }
    }
    const double n = (double)p.taps;
#pragma unroll
    for (int j = 0; j < kPx; ++j) {
        if (ox + j >= p.nx) break;
        const double s1 = (double)st[j] + (double)sf[j] * (1.0 / 65536.0);
        const size_t o = (size_t)(oy - p.out_row0) * p.nx + ox + j;
        if (WANT_TPI) {
            const double xs = (double)p.in[(size_t)(oy - p.in_row0) * p.nx + ox + j];
            const int cy = oy + p.centre, cx = ox + j + p.centre;
            double x_ctr = xs;
            if (p.centre != 0)
                x_ctr = (cy >= 0 && cy < p.gny && cx >= 0 && cx < p.nx)
                            ? (double)p.in[(size_t)(cy - p.in_row0) * p.nx + cx] : 0.0;
            p.tpi[o] = (float)(xs - (s1 - x_ctr) / (n - 1.0));
        }
        if (WANT_STD) {
            double var = (st2[j] - s1 * s1 / n) / (n - 1.0);
            if (var < 0.0) var = 0.0;
            p.sd[o] = (float)sqrt(var);
        }
    }
}

template <bool WANT_TPI, bool WANT_STD>
__global__ __launch_bounds__(kThreads) void big_disc_kernel(BigArgs p) {
    const int ox = blockIdx.x * kThreads + threadIdx.x;
    const int oy = p.out_row0 + blockIdx.y;
    if (ox >= p.nx) return;
    // every term is an exact integer in float64 (the planes are; a difference of two entries of a column is the sum of a
    // run, far below 2^53), and so are the sums over the disc's columns up to 2^53
    double st = 0.0, sf = 0.0, st2 = 0.0, st2b = 0.0, sc = 0.0;
    for (int k = 0; k < p.size; ++k) {
        const unsigned packed = (unsigned)p.runs[k];  // wave-uniform
        const int lo = (int)(packed & 0xffffu) - 32768, hi = (int)(packed >> 16) - 32768;
        const int x = ox + p.off_min + k;
        if (x < 0 || x >= p.nx) continue;  // zero padding: columns outside contribute nothing
        // rows [oy+lo, oy+hi] clipped to the DEM; the planes cover every in-DEM row needed
        const int top = min(max(oy + hi + 1, 0), p.gny) - p.p_row0;
        const int bot = min(max(oy + lo, 0), p.gny) - p.p_row0;
        if (top <= bot) continue;
        const size_t a = (size_t)top * p.nx + x, b = (size_t)bot * p.nx + x;
        st += p.qt[a] - p.qt[b];
        sf += p.qf[a] - p.qf[b];
        sc += p.qc[a] - p.qc[b];
        if (WANT_STD) {
            st2 += p.qt2[a] - p.qt2[b];
            st2b += p.qt2b[a] - p.qt2b[b];
        }
    }
    const double n = (double)p.taps;
    const double s1 = st + sf * (1.0 / 65536.0);
    const size_t o = (size_t)(oy - p.out_row0) * p.nx + ox;
    const bool missing = sc > 0.0;
    const float nan = __uint_as_float(0x7fc00000u);
    if (WANT_TPI) {
        const double xs = (double)p.in[(size_t)(oy - p.in_row0) * p.nx + ox];
        const int cy = oy + p.centre, cx = ox + p.centre;
        double x_ctr = xs;
        if (p.centre != 0)
            x_ctr = (cy >= 0 && cy < p.gny && cx >= 0 && cx < p.nx)
                        ? (double)p.in[(size_t)(cy - p.in_row0) * p.nx + cx] : 0.0;
        p.tpi[o] = missing ? nan : (float)(xs - (s1 - x_ctr) / (n - 1.0));
    }
    if (WANT_STD) {
        const double s2 = st2 * 16777216.0 + st2b;  // exact while it stays below 2^53
        double var;
        if (st2 < 268435456.0) {  // sum trunc(x)^2 below 2^52 + 2^46: the expression of the narrow planes, on the same exact sums
            var = (s2 - s1 * s1 / n) / (n - 1.0);
        } else {
            // 2^32 (n s2 - s1^2) with s1 = T + Sg / 2^16, exactly, in 128 bits (disc_wave_impl.hpp, std_from_exact_sums)
            const __int128 T = (__int128)(long long)st, Sg = (__int128)(long long)sf;
            const __int128 S2 = (((__int128)(long long)st2) << 24) + (__int128)(long long)st2b;
            const __int128 A = (__int128)p.taps * S2 - T * T;
            const __int128 num = (A << 32) - ((T * Sg) << 17) - Sg * Sg;
            const bool neg = num < 0;
            const unsigned __int128 mag = neg ? (unsigned __int128)(-num) : (unsigned __int128)num;
            const double d = (double)(unsigned long long)(mag >> 64) * 18446744073709551616.0 + (double)(unsigned long long)mag;
            var = (neg ? -d : d) * (1.0 / 4294967296.0) / (n * (n - 1.0));
        }
        if (var < 0.0) var = 0.0;
        p.sd[o] = missing ? nan : (float)sqrt(var);
    }
}

}  // namespace

int launch_disc_big(const Block& b, const DiscRuns& disc, float* tpi_out, float* std_out) {
    Context& c = ctx();
    TOPO_REQUIRE(disc.size <= 32767, "disc size %d too large", disc.size);
    BigArgs a{};
    a.in = b.in;
    a.in_row0 = b.in_row0;
    a.gny = b.gny;
    a.nx = b.nx;
    a.out_row0 = b.out_row0;
    a.out_rows = b.out_rows;
    a.p_row0 = std::max(0, b.out_row0 + disc.dj_min);
    const int p_end = std::min(b.gny, b.out_row0 + b.out_rows + disc.dj_max);
    a.p_rows = p_end - a.p_row0;
    a.off_min = disc.di_min;
    a.size = disc.di_max - disc.di_min + 1;
    a.centre = disc.centre_dj;
    a.taps = disc.taps;
    a.tpi = tpi_out;
    a.sd = std_out;
    const size_t plane = (size_t)(a.p_rows + 1) * b.nx * sizeof(double);
    void *q0 = nullptr, *q1 = nullptr, *q2 = nullptr, *i0 = nullptr, *i1 = nullptr, *fl = nullptr;
    if (std_out) {
        TOPO_TRY(workspace(6, plane, &q2));
        a.qt2 = (double*)q2;
    }
    TOPO_TRY(workspace(1, plane / 2, &i0));
    TOPO_TRY(workspace(2, plane / 2, &i1));
    TOPO_TRY(workspace(0, 64, &fl));
    a.it = (uint32_t*)i0;
    a.ifr = (uint32_t*)i1;
    a.flags = (int*)fl;
    // the mask is symmetric under transposition: the per-row runs are also the per-column runs
    std::vector<int> packed(a.size);
    for (int k = 0; k < a.size; ++k) packed[k] = (int)(((unsigned)((int)disc.lo[k] + 32768)) | ((unsigned)((int)disc.hi[k] + 32768) << 16));
    void* d_runs = nullptr;
    TOPO_TRY(upload_table(0, packed.data(), packed.size() * sizeof(int), &d_runs));
    a.runs = (const int*)d_runs;
    TOPO_TRY(check_grid_rows(std::max(b.out_rows, b.in_rows), "disc (prefix planes)"));
    const dim3 pgrid((b.nx + kThreads - 1) / kThreads), ggrid((b.nx + kThreads - 1) / kThreads, b.out_rows);

    // narrow planes first; which gather follows depends on what the prefix pass saw in the block
    TOPO_HIP(hipMemsetAsync(a.flags, 0, sizeof(int), c.compute));
    const int nseg = (a.p_rows + kSeg - 1) / kSeg;
    void* segs = nullptr;
    TOPO_TRY(workspace(7, (size_t)nseg * b.nx * 16, &segs));
    a.seg_t = (uint32_t*)segs;
    a.seg_f = a.seg_t + (size_t)nseg * b.nx;
    a.seg_t2 = (double*)(a.seg_f + (size_t)nseg * b.nx);
    hipLaunchKernelGGL(big_prefix_int_kernel, dim3(pgrid.x, nseg), dim3(kThreads), 0, c.compute, a);
    if (nseg > 1) {
        hipLaunchKernelGGL(big_segment_scan_kernel, pgrid, dim3(kThreads), 0, c.compute, a, nseg);
        hipLaunchKernelGGL(big_segment_add_kernel, dim3(pgrid.x, a.p_rows - kSeg), dim3(kThreads), 0, c.compute, a);
    }
    TOPO_HIP(hipGetLastError());
    int flags = 0;
    TOPO_HIP(hipMemcpyAsync(&flags, a.flags, sizeof(int), hipMemcpyDeviceToHost, c.compute));
    TOPO_HIP(hipStreamSynchronize(c.compute));
    if (!(flags & kBigBad)) {
        const bool frac = (flags & kBigFrac) != 0;
        const dim3 igrid((b.nx + kThreads * kPx - 1) / (kThreads * kPx), b.out_rows);
        auto go = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, igrid, dim3(kThreads), 0, c.compute, a);
        };
        if (tpi_out && std_out) frac ? go(big_disc_int_kernel<true, true, true>) : go(big_disc_int_kernel<true, true, false>);
        else if (tpi_out) frac ? go(big_disc_int_kernel<true, false, true>) : go(big_disc_int_kernel<true, false, false>);
        else frac ? go(big_disc_int_kernel<false, true, true>) : go(big_disc_int_kernel<false, true, false>);
        TOPO_HIP(hipGetLastError());
        return TOPO_AMD_OK;
    }
    // non-finite or absurd samples: float64 planes, so that NaN propagates and nothing wraps
    TOPO_TRY(workspace(4, 3 * plane, &q0));
    TOPO_TRY(workspace(5, plane, &q1));
    a.qt = (double*)q0;
    a.qt2b = (double*)((char*)q0 + plane);
    a.qc = (double*)((char*)q0 + 2 * plane);
    a.qf = (double*)q1;

    hipLaunchKernelGGL(big_prefix_kernel, dim3((b.nx + kThreads - 1) / kThreads), dim3(kThreads), 0,
                       c.compute, a);
    TOPO_HIP(hipGetLastError());
    TOPO_TRY(check_grid_rows(b.out_rows, "disc (float64 prefix planes)"));
    dim3 grid((b.nx + kThreads - 1) / kThreads, b.out_rows);
    if (tpi_out && std_out) hipLaunchKernelGGL((big_disc_kernel<true, true>), grid, dim3(kThreads), 0, c.compute, a);
    else if (tpi_out) hipLaunchKernelGGL((big_disc_kernel<true, false>), grid, dim3(kThreads), 0, c.compute, a);
    else hipLaunchKernelGGL((big_disc_kernel<false, true>), grid, dim3(kThreads), 0, c.compute, a);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

}  // namespace topo
