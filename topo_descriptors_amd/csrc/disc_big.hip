// K1/K2 for discs too large for an LDS tile (more than ~120 px across): column prefix sums kept
// in HBM as float64, one prefix difference per column offset per pixel.
//
// The reference handles any size with the same FFT (topo.py:175); its example script goes up
// to 100 km scales, i.e. discs of ~2000 px (scripts/compute_topo_descriptors.py:25-38).  This
// path is the completeness fallback: O(size) reads per pixel from L2/HBM instead of O(1) LDS
// traffic, but exact - the prefix sums of trunc(x) and trunc(x)^2 are integers in float64, the
// fractional parts are summed separately - and it reproduces the zero padding by clamping the
// prefix rows at the DEM edge and skipping columns outside it.
#include "common.hpp"

namespace topo {

namespace {

constexpr int kThreads = 256;

struct BigArgs {
    const float* in;
    double* qt;   // prefix over rows of trunc(x),   (rows + 1) x nx
    double* qf;   // prefix of x - trunc(x)
    double* qt2;  // prefix of trunc(x)^2 (only when STD is wanted)
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
    double st = 0.0, sf = 0.0, st2 = 0.0;
    p.qt[x] = 0.0;
    p.qf[x] = 0.0;
    if (p.qt2) p.qt2[x] = 0.0;
    const float* src = p.in + (size_t)(p.p_row0 - p.in_row0) * p.nx + x;
    for (int r = 0; r < p.p_rows; ++r) {
        const float v = src[(size_t)r * p.nx];
        const float t = truncf(v);
        st += (double)t;
        sf += (double)(v - t);
        const size_t o = (size_t)(r + 1) * p.nx + x;
        p.qt[o] = st;
        p.qf[o] = sf;
        if (p.qt2) {
            st2 += (double)t * (double)t;
            p.qt2[o] = st2;
        }
    }
}

template <bool WANT_TPI, bool WANT_STD>
__global__ __launch_bounds__(kThreads) void big_disc_kernel(BigArgs p) {
    const int ox = blockIdx.x * kThreads + threadIdx.x;
    const int oy = p.out_row0 + blockIdx.y;
    if (ox >= p.nx) return;
    double st = 0.0, sf = 0.0, st2 = 0.0;
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
        if (WANT_STD) st2 += p.qt2[a] - p.qt2[b];
    }
    const double n = (double)p.taps;
    const double s1 = st + sf;
    const size_t o = (size_t)(oy - p.out_row0) * p.nx + ox;
    if (WANT_TPI) {
        const double xs = (double)p.in[(size_t)(oy - p.in_row0) * p.nx + ox];
        const int cy = oy + p.centre, cx = ox + p.centre;
        double x_ctr = xs;
        if (p.centre != 0)
            x_ctr = (cy >= 0 && cy < p.gny && cx >= 0 && cx < p.nx)
                        ? (double)p.in[(size_t)(cy - p.in_row0) * p.nx + cx] : 0.0;
        p.tpi[o] = (float)(xs - (s1 - x_ctr) / (n - 1.0));
    }
    if (WANT_STD) {
        double var = (st2 - s1 * s1 / n) / (n - 1.0);
        if (var < 0.0) var = 0.0;  // keeps NaN, like np.clip
        p.sd[o] = (float)sqrt(var);
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
    void *q0 = nullptr, *q1 = nullptr, *q2 = nullptr;
    TOPO_TRY(workspace(4, plane, &q0));
    TOPO_TRY(workspace(5, plane, &q1));
    a.qt = (double*)q0;
    a.qf = (double*)q1;
    if (std_out) {
        TOPO_TRY(workspace(6, plane, &q2));
        a.qt2 = (double*)q2;
    }
    // the mask is symmetric under transposition: the per-row runs are also the per-column runs
    std::vector<int> packed(a.size);
    for (int k = 0; k < a.size; ++k) packed[k] = (int)(((unsigned)((int)disc.lo[k] + 32768)) | ((unsigned)((int)disc.hi[k] + 32768) << 16));
    void* d_runs = nullptr;
    TOPO_TRY(upload_table(0, packed.data(), packed.size() * sizeof(int), &d_runs));
    a.runs = (const int*)d_runs;

    hipLaunchKernelGGL(big_prefix_kernel, dim3((b.nx + kThreads - 1) / kThreads), dim3(kThreads), 0,
                       c.compute, a);
    TOPO_HIP(hipGetLastError());
    dim3 grid((b.nx + kThreads - 1) / kThreads, b.out_rows);
    if (tpi_out && std_out) hipLaunchKernelGGL((big_disc_kernel<true, true>), grid, dim3(kThreads), 0, c.compute, a);
    else if (tpi_out) hipLaunchKernelGGL((big_disc_kernel<true, false>), grid, dim3(kThreads), 0, c.compute, a);
    else hipLaunchKernelGGL((big_disc_kernel<false, true>), grid, dim3(kThreads), 0, c.compute, a);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

}  // namespace topo
