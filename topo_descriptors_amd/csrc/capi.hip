// C ABI of libtopo_amd.so (declared in include/topo_amd.h): context, memory, the row-block
// and host-buffer descriptor entry points, and RCCL ghost-row exchange for row shards.
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "common.hpp"

#include <sys/mman.h>

#include <thread>

namespace topo {

static thread_local std::string g_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
}

Context& ctx() {
    static Context c;
    return c;
}

std::recursive_mutex& call_mutex() {
    static std::recursive_mutex mu;
    return mu;
}
namespace {
thread_local int t_bound_device = -1;  // the device this thread's HIP calls go to, as far as the library has set it
}
CallGuard::CallGuard() : lock(call_mutex()) {
    const Context& c = ctx();
    if (c.ready && t_bound_device != c.device) {
        if (hipSetDevice(c.device) == hipSuccess) t_bound_device = c.device;
    }
}

int require_ready() {
    if (!ctx().ready) {
        set_error("libtopo_amd: call topo_amd_init(device) first");
        return TOPO_AMD_ENODEV;
    }
    return TOPO_AMD_OK;
}

int workspace(int slot, size_t bytes, void** out) {
    Context& c = ctx();
    if (c.ws_bytes[slot] < bytes) {
        if (c.ws[slot]) {
            TOPO_HIP(hipStreamSynchronize(c.compute));
            TOPO_HIP(hipFree(c.ws[slot]));
            c.ws[slot] = nullptr;
            c.ws_bytes[slot] = 0;
        }
        TOPO_HIP(hipMalloc(&c.ws[slot], bytes));
        c.ws_bytes[slot] = bytes;
        dem_memo_forget(c.ws[slot], bytes);  // (a recycled address)
    }
    *out = c.ws[slot];
    return TOPO_AMD_OK;
}

// ---- dem_memo and the raster class (common.hpp) -------------------------------------------------------------------------
namespace {
struct DemMemo {
    const void* in = nullptr;
    int rows = 0, nx = 0;
    unsigned long used = 0;
    unsigned asked = 0;     // calls that asked "wild?" (every 32nd one is told no: the fused kernel looks again; time only)
    bool cls_valid = false;  // the block is a whole raster and has been scanned
    RasterClass cls;
};
constexpr int kMemos = 8;
constexpr int kMemoWords = 4;
DemMemo g_memo[kMemos];
uint32_t* g_memo_words = nullptr;  // pinned: kMemoWords words per entry: tiles, fractional tiles, wild sample seen
unsigned long g_memo_clock = 0;
std::mutex g_memo_mu;              // (several driver threads: topo_amd_shard_layout is per thread for them)

// g_memo_mu held
int memo_slot(const Block& b, bool create) {
    if (!g_memo_words) {
        if (hipHostMalloc((void**)&g_memo_words, kMemos * kMemoWords * sizeof(uint32_t), hipHostMallocMapped) != hipSuccess) return -1;
        std::memset(g_memo_words, 0, kMemos * kMemoWords * sizeof(uint32_t));
    }
    int oldest = 0;
    for (int k = 0; k < kMemos; ++k) {
        if (g_memo[k].in == b.in && g_memo[k].rows == b.in_rows && g_memo[k].nx == b.nx) {
            g_memo[k].used = ++g_memo_clock;
            return k;
        }
        if (g_memo[k].used < g_memo[oldest].used) oldest = k;
    }
    if (!create) return -1;
    g_memo[oldest] = DemMemo{b.in, b.in_rows, b.nx, ++g_memo_clock, 0, false, RasterClass()};
    // (a launch in flight may still write the evicted entry's words: they are cleared here, and a late report for
    // another DEM can at worst pick the slower first kernel once - the results do not depend on that choice)
    for (int w = 0; w < kMemoWords; ++w) g_memo_words[kMemoWords * oldest + w] = 0;
    return oldest;
}
// What was declared for partial row blocks (topo_amd_raster_class_set / _from_scan / topo_amd_shard_classify): keyed by the
// device rows the declaration was made for and the shape of the raster they belong to.  A call's block finds its class
// when its first row lies inside a declared range of a raster of the same shape.  Dropped like the memo: whenever the
// library writes or frees memory that overlaps the range, by topo_amd_dem_changed, and by a withdrawal.  Nothing here
// belongs to a thread or outlives its memory: two rasters in one process cannot inherit each other's class.
struct Declared {
    uintptr_t lo = 0, hi = 0;  // [lo, hi): the device rows
    int gny = 0, nx = 0;
    unsigned long used = 0;
    RasterClass cls;
};
constexpr size_t kMaxDeclared = 256;
std::vector<Declared> g_declared;  // g_memo_mu held
// g_memo_mu held
const Declared* find_declared(const Block& b) {
    const uintptr_t p = (uintptr_t)b.in;
    for (Declared& e : g_declared)
        if (p >= e.lo && p < e.hi && e.gny == b.gny && e.nx == b.nx) {
            e.used = ++g_memo_clock;
            return &e;
        }
    return nullptr;
}
bool declared_class(const Block& b, RasterClass* out) {
    std::lock_guard<std::mutex> lock(g_memo_mu);
    const Declared* e = find_declared(b);
    if (e && out) *out = e->cls;
    return e != nullptr;
}
void declare_class(const float* block, int in_rows, int gny, int nx, const RasterClass& cls) {
    std::lock_guard<std::mutex> lock(g_memo_mu);
    const uintptr_t lo = (uintptr_t)block, hi = lo + (size_t)in_rows * nx * sizeof(float);
    // a declaration replaces whatever was declared for overlapping memory
    g_declared.erase(std::remove_if(g_declared.begin(), g_declared.end(), [&](const Declared& e) { return e.lo < hi && lo < e.hi; }),
                     g_declared.end());
    if (g_declared.size() >= kMaxDeclared) {
        auto oldest = std::min_element(g_declared.begin(), g_declared.end(), [](const Declared& x, const Declared& y) { return x.used < y.used; });
        g_declared.erase(oldest);
    }
    Declared e;
    e.lo = lo;
    e.hi = hi;
    e.gny = gny;
    e.nx = nx;
    e.used = ++g_memo_clock;
    e.cls = cls;
    g_declared.push_back(e);
}
// g_memo_mu held
void forget_declared(uintptr_t lo, uintptr_t hi) {
    g_declared.erase(std::remove_if(g_declared.begin(), g_declared.end(), [&](const Declared& e) { return e.lo < hi && lo < e.hi; }),
                     g_declared.end());
}
}  // namespace

uint32_t* dem_memo_report(const Block& b) {
    std::lock_guard<std::mutex> lock(g_memo_mu);
    const int k = memo_slot(b, true);
    return k < 0 ? nullptr : g_memo_words + kMemoWords * k;
}
bool dem_memo_mostly_fractional(const Block& b) {
    std::lock_guard<std::mutex> lock(g_memo_mu);
    const int k = memo_slot(b, false);
    if (k < 0) return false;
    // (both routes report afresh on every call, so a wrong guess costs one slower call and corrects itself)
    const uint32_t tiles = *(volatile uint32_t*)(g_memo_words + kMemoWords * k), frac = *(volatile uint32_t*)(g_memo_words + kMemoWords * k + 1);
    return tiles > 0 && 2 * frac > tiles;
}

uint32_t* dem_memo_wild_word(const Block& b) {
    std::lock_guard<std::mutex> lock(g_memo_mu);
    const int k = memo_slot(b, true);
    return k < 0 ? nullptr : g_memo_words + kMemoWords * k + 2;
}
bool dem_memo_wild(const Block& b) {
    std::lock_guard<std::mutex> lock(g_memo_mu);
    const int k = memo_slot(b, false);
    if (k < 0) return false;
    if (++g_memo[k].asked % 32 == 0) {  // (a caller may have rewritten the buffer with kernels of its own and not said so)
        *(volatile uint32_t*)(g_memo_words + kMemoWords * k + 2) = 0;
        return false;
    }
    return *(volatile uint32_t*)(g_memo_words + kMemoWords * k + 2) != 0;
}
// every entry whose block overlaps [p, p + bytes) (bytes == 0: that starts at p) is dropped
void dem_memo_forget(const void* p, size_t bytes) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_memo_mu);
    const uintptr_t lo = (uintptr_t)p, hi = lo + (bytes ? bytes : 1);
    forget_declared(lo, hi);
    for (int k = 0; k < kMemos; ++k) {
        DemMemo& m = g_memo[k];
        if (!m.in) continue;
        const uintptr_t a = (uintptr_t)m.in, b = a + (size_t)m.rows * m.nx * sizeof(float);
        if (a < hi && lo < b) {
            m = DemMemo();
            if (g_memo_words)
                for (int w = 0; w < kMemoWords; ++w) g_memo_words[kMemoWords * k + w] = 0;
        }
    }
}

// ---- lattice scan of a raster (the raster class) ---------------------------------------------------------------------------
// The lattice belongs to the GLOBAL grid: rows step_r / 2 + i step_r, columns step_c / 2 + j step_c with step = extent / 128
// (about 16 K points), so the scans of the row blocks of a raster add up to the scan of the whole raster, point for point.
namespace {
constexpr float kClassLarge = 1.0e5f;     // kWild of gauss.hip: what the f16 matrix-core kernels stage as 0 and repair
constexpr float kClassOrdinary = 262144.0f;  // kAbsLim of the disc kernels
struct Scan {
    unsigned long long taken = 0, large = 0, frac = 0;
    float lo = INFINITY, hi = -INFINITY;
    void add(float x) {
        ++taken;
        const float a = std::fabs(x);
        if (a > kClassLarge && a <= 3.0e38f) ++large;
        if (a <= 3.0e38f && x != std::trunc(x)) ++frac;
        if (a <= kClassOrdinary) {  // (false for NaN)
            lo = std::min(lo, x);
            hi = std::max(hi, x);
        }
    }
};
inline int lattice_step(int extent) { return std::max(1, extent / 128); }
RasterClass class_of(const Scan& s) {
    RasterClass c;
    c.large = 4 * s.large > s.taken;
    c.frac_share = s.taken ? (float)((double)s.frac / (double)s.taken) : 0.0f;
    c.lo = s.lo;
    c.hi = s.hi;
    return c;
}
__device__ __forceinline__ uint32_t ordered_bits(float f) {  // unsigned order = float order (finite values)
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
float from_ordered_bits(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    std::memcpy(&f, &u, sizeof(f));
    return f;
}
// words: [0] taken, [1] large, [2] min key, [3] max key, [4] fractional (pinned host memory)
__global__ __launch_bounds__(256) void raster_scan_kernel(const float* in, int in_rows, int in_row0, int nx, int own_row0, int own_rows,
                                                          int step_r, int step_c, int ni, int nj, uint32_t* words) {
    const int idx = (int)(blockIdx.x * 256 + threadIdx.x);
    const int i = idx / nj, j = idx - i * nj;
    const int r = step_r / 2 + i * step_r, c = step_c / 2 + j * step_c;
    const bool take = i < ni && r >= own_row0 && r < own_row0 + own_rows && r >= in_row0 && r < in_row0 + in_rows && c < nx;
    const float x = take ? in[(size_t)(r - in_row0) * nx + c] : 0.0f;
    const float a = fabsf(x);
    const bool large = take && a > kClassLarge && a <= 3.0e38f;
    const bool ordinary = take && a <= kClassOrdinary;
    const bool fractional = take && a <= 3.0e38f && x != truncf(x);
    uint32_t kmin = ordinary ? ordered_bits(x) : 0xffffffffu, kmax = ordinary ? ordered_bits(x) : 0u;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, m));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, m));
    }
    const unsigned long long mt = __builtin_amdgcn_ballot_w64(take), ml = __builtin_amdgcn_ballot_w64(large);
    const unsigned long long mf = __builtin_amdgcn_ballot_w64(fractional);
    if ((threadIdx.x & 63) == 0 && mt) {
        __hip_atomic_fetch_add(words, (uint32_t)__builtin_popcountll(mt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (ml) __hip_atomic_fetch_add(words + 1, (uint32_t)__builtin_popcountll(ml), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (mf) __hip_atomic_fetch_add(words + 4, (uint32_t)__builtin_popcountll(mf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_min(words + 2, kmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_max(words + 3, kmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
uint32_t* g_scan_words = nullptr;  // pinned
std::mutex g_scan_mu;
// scans the lattice points in rows [own_row0, own_row0 + own_rows) of the block; synchronises the compute stream
int scan_block(const Block& b, int own_row0, int own_rows, Scan* out) {
    std::lock_guard<std::mutex> lock(g_scan_mu);
    Context& c = ctx();
    if (!g_scan_words) TOPO_HIP(hipHostMalloc((void**)&g_scan_words, 8 * sizeof(uint32_t), hipHostMallocMapped));
    TOPO_HIP(hipStreamSynchronize(c.compute));  // (no scan of an earlier call in flight on the words; the block's data are final)
    g_scan_words[0] = g_scan_words[1] = g_scan_words[4] = 0;
    g_scan_words[2] = 0xffffffffu;
    g_scan_words[3] = 0u;
    const int step_r = lattice_step(b.gny), step_c = lattice_step(b.nx);
    const int ni = (b.gny - step_r / 2 + step_r - 1) / step_r, nj = (b.nx - step_c / 2 + step_c - 1) / step_c;
    const long n = (long)ni * nj;
    hipLaunchKernelGGL(raster_scan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.compute, b.in, b.in_rows, b.in_row0, b.nx,
                       own_row0, own_rows, step_r, step_c, ni, nj, g_scan_words);
    TOPO_HIP(hipGetLastError());
    TOPO_HIP(hipStreamSynchronize(c.compute));
    out->taken += g_scan_words[0];
    out->large += g_scan_words[1];
    out->frac += g_scan_words[4];
    if (g_scan_words[2] <= g_scan_words[3]) {
        out->lo = std::min(out->lo, from_ordered_bits(g_scan_words[2]));
        out->hi = std::max(out->hi, from_ordered_bits(g_scan_words[3]));
    }
    return TOPO_AMD_OK;
}
// the same lattice on a host array (the host-buffer entry points: no launch, no synchronisation)
RasterClass scan_host(const float* dem, int ny, int nx) {
    Scan s;
    const int step_r = lattice_step(ny), step_c = lattice_step(nx);
    for (int r = step_r / 2; r < ny; r += step_r)
        for (int c = step_c / 2; c < nx; c += step_c) s.add(dem[(size_t)r * nx + c]);
    return class_of(s);
}

// The class of the call in flight on this thread: set by the outermost entry point, resolved when a launcher first asks.
thread_local const Block* t_call_block = nullptr;  // the block the outermost entry point was given
thread_local bool t_class_known = false;
thread_local RasterClass t_class;
}  // namespace

RasterClass current_class() {
    if (t_class_known) return t_class;
    RasterClass c;  // nothing declared: an ordinary DEM in whole metres
    const Block* b = t_call_block;
    if (b != nullptr && !(b->in_row0 == 0 && b->in_rows == b->gny)) (void)declared_class(*b, &c);
    if (b != nullptr && b->in_row0 == 0 && b->in_rows == b->gny) {
        // the block IS the raster: its own scan, remembered with the block
        bool have = false;
        {
            std::lock_guard<std::mutex> lock(g_memo_mu);
            const int k = memo_slot(*b, true);
            if (k >= 0 && g_memo[k].cls_valid) {
                c = g_memo[k].cls;
                have = true;
            }
        }
        if (!have) {
            Scan s;
            if (scan_block(*b, 0, b->gny, &s) == TOPO_AMD_OK) {
                c = class_of(s);
                std::lock_guard<std::mutex> lock(g_memo_mu);
                const int k = memo_slot(*b, true);
                if (k >= 0) {
                    g_memo[k].cls = c;
                    g_memo[k].cls_valid = true;
                }
            }
        }
    }
    t_class = c;
    t_class_known = true;
    return c;
}

namespace {
// Around the launchers of one entry point.  The outermost scope on a thread names the block (or, for the host-buffer
// entry points, the class found on the caller's array); scopes inside it - the device entry points the host-buffer ones
// call, the Gaussian in front of a disc - change nothing, so a smoothed plane inherits the class of the DEM it came from.
struct ClassScope {
    bool outer;
    explicit ClassScope(const Block& b) : outer(t_call_block == nullptr && !t_class_known) {
        if (outer) t_call_block = &b;
    }
    explicit ClassScope(const RasterClass& c) : outer(t_call_block == nullptr && !t_class_known) {
        if (outer) {
            t_class = c;
            t_class_known = true;
        }
    }
    ~ClassScope() {
        if (outer) {
            t_call_block = nullptr;
            t_class_known = false;
        }
    }
    ClassScope(const ClassScope&) = delete;
    ClassScope& operator=(const ClassScope&) = delete;
};
inline void forget_plane(const void* p, int rows, int nx) {
    if (p) dem_memo_forget(p, (size_t)rows * nx * sizeof(float));
}
}  // namespace

// Small parameter tables: pinned staging + async copy on the compute stream.  The previous
// content is remembered so a loop over the same parameters uploads nothing.
namespace {
struct TableSlot {
    void* pinned = nullptr;
    size_t cap = 0;
    std::vector<char> last;
    hipEvent_t copied = nullptr;
};
TableSlot g_slots[6];
}  // namespace

int upload_table(int slot, const void* host, size_t bytes, void** out) {
    Context& c = ctx();
    TableSlot& s = g_slots[slot];
    if (bytes == 0) bytes = 4;
    if (c.tab_bytes[slot] < bytes || s.cap < bytes) {
        const size_t cap = bytes * 2 + 256;
        TOPO_HIP(hipStreamSynchronize(c.compute));
        if (c.tab[slot]) TOPO_HIP(hipFree(c.tab[slot]));
        if (s.pinned) TOPO_HIP(hipHostFree(s.pinned));
        TOPO_HIP(hipMalloc(&c.tab[slot], cap));
        TOPO_HIP(hipHostMalloc(&s.pinned, cap, hipHostMallocDefault));
        c.tab_bytes[slot] = cap;
        s.cap = cap;
        s.last.clear();
        if (!s.copied) TOPO_HIP(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
    }
    if (s.last.size() != bytes || std::memcmp(s.last.data(), host, bytes) != 0) {
        if (!s.last.empty()) TOPO_HIP(hipEventSynchronize(s.copied));  // staging still in use?
        std::memcpy(s.pinned, host, bytes);
        TOPO_HIP(hipMemcpyAsync(c.tab[slot], s.pinned, bytes, hipMemcpyHostToDevice, c.compute));
        TOPO_HIP(hipEventRecord(s.copied, c.compute));
        s.last.assign((const char*)host, (const char*)host + bytes);
    }
    *out = c.tab[slot];
    return TOPO_AMD_OK;
}

int check_block(const Block& b, int need_above, int need_below, const char* who) {
    TOPO_REQUIRE(b.in != nullptr, "%s: input pointer is NULL", who);
    TOPO_REQUIRE(b.gny >= 1 && b.nx >= 1, "%s: empty DEM %d x %d", who, b.gny, b.nx);
    TOPO_REQUIRE(b.in_rows >= 1 && b.in_row0 >= 0 && b.in_row0 + b.in_rows <= b.gny,
                 "%s: block rows [%d, %d) outside the DEM of %d rows", who, b.in_row0,
                 b.in_row0 + b.in_rows, b.gny);
    TOPO_REQUIRE(b.out_rows >= 1 && b.out_row0 >= 0 && b.out_row0 + b.out_rows <= b.gny,
                 "%s: output rows [%d, %d) outside the DEM of %d rows", who, b.out_row0,
                 b.out_row0 + b.out_rows, b.gny);
    const int first = std::max(0, b.out_row0 - need_above);
    const int last = std::min(b.gny, b.out_row0 + b.out_rows + need_below);
    TOPO_REQUIRE(b.in_row0 <= first && b.in_row0 + b.in_rows >= last,
                 "%s: block rows [%d, %d) do not cover the %d/%d ghost rows needed by output rows "
                 "[%d, %d)", who, b.in_row0, b.in_row0 + b.in_rows, need_above, need_below,
                 b.out_row0, b.out_row0 + b.out_rows);
    return TOPO_AMD_OK;
}

// ---- RCCL state -----------------------------------------------------------------------------
namespace {
struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1;
    bool halo_pending = false;
} g_comm;
// declared ghost depth of the shard buffers (topo_amd_shard_layout): per calling thread, so that two threads (or an
// application and a ShardedDEM inside it) that drive shards with different buffer layouts do not see each other's
// shards.  A thread that has never declared a layout sees the process-wide one - the last one any thread declared (an
// application that declares it once at set-up and drives the shards from worker threads: ADVICE r03).
thread_local int t_layout_above = -1, t_layout_below = -1;
thread_local bool t_layout_set = false;
std::atomic<int> g_layout_above{-1}, g_layout_below{-1};
int layout_above() { return t_layout_set ? t_layout_above : g_layout_above.load(); }
int layout_below() { return t_layout_set ? t_layout_below : g_layout_below.load(); }

// TOPO_AMD_HALO_LOOPBACK=1 with a communicator of one rank: the exchange talks to itself
bool halo_loopback() {
    const char* e = std::getenv("TOPO_AMD_HALO_LOOPBACK");
    return e && e[0] == '1' && g_comm.size == 1 && g_comm.comm != nullptr;
}

#define TOPO_NCCL(call)                                                                    \
    do {                                                                                   \
        ncclResult_t r_ = (call);                                                          \
        if (r_ != ncclSuccess) {                                                           \
            ::topo::set_error("%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_),      \
                              __FILE__, __LINE__);                                         \
            return TOPO_AMD_ERCCL;                                                         \
        }                                                                                  \
    } while (0)

// compose pre-smoothing + disc: TPI/STD with sigma (topo.py:172-173, :297-298)
int tpi_std_block(const Block& b, int size, double sigma, float* tpi_out, float* std_out) {
    DiscRuns disc;
    TOPO_TRY(build_disc(size, &disc));
    const int above = -disc.dj_min, below = disc.dj_max;
    if (!(sigma > 0.0)) {
        TOPO_TRY(check_block(b, above, below, "tpi_std"));
        return launch_tpi_std(b, disc, tpi_out, std_out);
    }
    const int R = gaussian_radius(sigma);
    TOPO_TRY(check_block(b, above + R, below + R, "tpi_std(sigma)"));
    // smooth exactly the rows the disc will read, then run the disc on that plane
    const int s0 = std::max(0, b.out_row0 - above);
    const int s1 = std::min(b.gny, b.out_row0 + b.out_rows + below);
    void* plane = nullptr;
    TOPO_TRY(workspace(3, (size_t)(s1 - s0) * b.nx * sizeof(float), &plane));
    Block g = b;
    g.out_row0 = s0;
    g.out_rows = s1 - s0;
    TOPO_TRY(launch_gaussian(g, sigma, sigma, (float*)plane));
    dem_memo_forget(plane, (size_t)(s1 - s0) * b.nx * sizeof(float));  // (the workspace holds another plane now)
    Block d = b;
    d.in = (const float*)plane;
    d.in_row0 = s0;
    d.in_rows = s1 - s0;
    return launch_tpi_std(d, disc, tpi_out, std_out);
}

// ghost rows the gradient cannot do without: the filter radius plus the row of the central difference
int gradient_halo(double sigma, double sig_ratio) {
    if (sigma <= 1.0) return 1;
    const double s_max = sig_ratio == 1.0 ? sigma : std::max(sigma, sigma * sig_ratio);
    return gaussian_radius(s_max) + 1;
}

// ghost rows a row SHARD of the gradient is laid out with and exchanges: what topo_amd_halo_rows(GRADIENT)
// answers and what topo_amd_shard_gradient uses (one function, so the two cannot drift apart).  Radii
// mfma_min_radius(true) .. 15 of the isotropic smooth ask for 17 rows: the matrix-core kernels take their
// accumulation offsets 16 rows into a 32-row tile (gauss.hip, mfma_rows_ok), and with R + 1 rows the interior and
// the seam strips of a shard would mix matrix-core and vector-ALU kernels depending on row0 % 32.
int gradient_shard_halo(double sigma, double sig_ratio) {
    if (sigma <= 1.0) return 1;
    if (sig_ratio == 0.0) sig_ratio = 1.0;
    if (sig_ratio != 1.0) return gradient_halo(sigma, sig_ratio);
    const int R = gaussian_radius(sigma);
    return (R >= mfma_min_radius(true) && R < 16) ? 17 : R + 1;
}

// RAII-less helper for the host-buffer entry points
// Device buffers of one host-buffer call, and the preparation of its result arrays.  A result array
// fresh from the allocator has no pages yet; faulting them in one by one under the download is what a
// host-buffer call spends most of its time on (tools/ubench/page_touch.cpp: 92 ms per GiB against 19 ms
// for the copy itself).  prefault() asks for huge pages and touches the array from a few threads
// while the upload and the kernels run; ready() joins them before the first download.
thread_local int t_valley_route = 0;  // the evaluation the calling thread's last valley / ridge call took (topo_amd_valley_route)
thread_local int t_host_chunks = 0;  // row chunks of the calling thread's last host-buffer call (topo_amd_host_chunks)
struct HostRun {
    std::vector<void*> bufs;
    std::vector<std::thread> touchers;
    explicit HostRun(bool entry_point = true) {
        if (entry_point) t_host_chunks = 0;
    }
    ~HostRun() {
        ready();
        for (size_t k = 0; k < bufs.size(); ++k) dem_memo_forget(bufs[k], sizes[k]);  // (the planes stay; what was known about their content goes)
    }
    std::vector<size_t> sizes;
    // the k-th plane this call asks for is the k-th plane of the context's pool (Context::host_planes), grown when it must be
    int alloc(void** p, size_t bytes) {
        Context& c = ctx();
        const size_t k = bufs.size();
        if (c.host_planes.size() <= k) {
            c.host_planes.resize(k + 1, nullptr);
            c.host_plane_bytes.resize(k + 1, 0);
        }
        if (c.host_plane_bytes[k] < bytes) {
            if (c.host_planes[k]) {
                TOPO_HIP(hipStreamSynchronize(c.compute));
                TOPO_HIP(hipFree(c.host_planes[k]));
                c.host_planes[k] = nullptr;
                c.host_plane_bytes[k] = 0;
            }
            TOPO_HIP(hipMalloc(&c.host_planes[k], bytes));
            c.host_plane_bytes[k] = bytes;
        }
        *p = c.host_planes[k];
        bufs.push_back(*p);
        sizes.push_back(bytes);
        dem_memo_forget(*p, c.host_plane_bytes[k]);  // (another call's data: nothing is known about this plane)
        return TOPO_AMD_OK;
    }
    void prefault(void* host, size_t bytes) {
        constexpr uintptr_t kPage = 4096, kHuge = (uintptr_t)2 << 20;
        static const bool enabled = [] {
            const char* e = std::getenv("TOPO_AMD_HOST_PREFAULT");  // 0 switches it off (for measurements)
            return !(e && e[0] == '0');
        }();
        if (!enabled || !host || bytes < 4 * kHuge) return;
        const uintptr_t lo = ((uintptr_t)host + kPage - 1) & ~(kPage - 1);
        const uintptr_t hi = ((uintptr_t)host + bytes) & ~(kPage - 1);
        if (hi <= lo) return;
        (void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);  // a hint; refused where THP is off
        const unsigned hw = std::thread::hardware_concurrency();
        const uintptr_t n = std::min<uintptr_t>(hw >= 16 ? 8 : (hw >= 4 ? 2 : 1), (hi - lo) / kHuge);
        const uintptr_t per = (((hi - lo) / n) + kHuge - 1) & ~(kHuge - 1);
        for (uintptr_t a = lo; a < hi; a += per) {
            const uintptr_t b = std::min(a + per, hi);
            try {
                touchers.emplace_back([a, b] {
                    // read and write back one byte per page: the array keeps whatever it held
                    for (uintptr_t q = a; q < b; q += kPage) {
                        volatile char* c = (volatile char*)q;
                        *c = *c;
                    }
                });
            } catch (...) {
                return;  // no thread to be had: the download faults the remaining pages in itself
            }
        }
    }
    void ready() {
        for (auto& t : touchers) t.join();
        touchers.clear();
    }
};

float* shift(float* p, int rows, int nx) { return p ? p + (size_t)rows * nx : nullptr; }

int download(void* host, const void* dev, size_t bytes) {
    if (!host) return TOPO_AMD_OK;
    TOPO_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx().compute));
    return TOPO_AMD_OK;
}

// ---- upload || kernels || download in row chunks -------------------------------------------------------
// A host-buffer call used to be three serial steps - 18.7 ms up, 1.3 ms of kernels, 19.9 ms down for TPI 67 px on 16384^2 -
// over a link that carries both directions at once (48.6 GB/s each way against 57 one way, profiles/r01_host_path.txt).  Row
// blocks give the single block's bits, so the call is cut into row chunks: the calling thread uploads chunk k on one
// stream and, as soon as the rows a chunk's outputs depend on are on the device, enqueues its kernels (on the block view
// "the rows uploaded so far") on the compute stream; a second host thread downloads the output rows of every finished
// chunk on a third stream (its own thread because copies to and from pageable memory block the caller).
// TOPO_AMD_HOST_PIPELINE=0: one chunk (the old order).  TOPO_AMD_HOST_CHUNK_MB: size of a chunk of the DEM (default 64).
// TOPO_AMD_HOST_DOWNLOADS=thread / inline: who issues the downloads (default: by the arrays, below).  All three are read at
// every call (a getenv each), so a test can walk through them in one process.
struct HostPlane {
    float* host;
    float* dev;
};
int pipeline_chunk_rows(int ny, int nx) {
    const char* e = std::getenv("TOPO_AMD_HOST_PIPELINE");
    if (e && *e == '0') return ny;
    e = std::getenv("TOPO_AMD_HOST_CHUNK_MB");
    const double chunk_mb = e && *e ? std::max(1.0, std::atof(e)) : 64.0;
    // whole tile rows of every kernel (60 and 64: 960), at least 960 rows
    long rows = (long)(chunk_mb * 1048576.0 / ((double)nx * sizeof(float)));
    rows = std::max(960L, rows / 960 * 960);
    return rows * 3 > ny ? ny : (int)rows;  // fewer than three chunks: nothing to overlap
}
// The cuts of a pipelined call: chunks of `chunk` rows, the last one half a chunk to a chunk and a half (no sliver at the end).
// (Round 6 tried a taper - a quarter and a half chunk at both ends, so that less stands in front of the first kernels and
// behind the last ones: 16384^2 TPI 67 px 26.05 -> 25.93 ms page-locked, 24.97 -> 24.80 pageable, profiles/r06_host_pipeline.txt:
// the fill and the drain are not what separates the call from the 22.4 ms of 1 GiB each way at 48 GB/s.  Not kept.)
std::vector<int> pipeline_cuts(int ny, int chunk) {
    std::vector<int> cut{0};
    int at = 0;
    while (ny - at > chunk + chunk / 2) cut.push_back(at += chunk);
    cut.push_back(ny);
    return cut;
}
// compute(view_rows, out_row0, out_rows): enqueue the kernels that write output rows [out_row0, out_row0 + out_rows) of
// every plane, reading rows [0, view_rows) of d_in.  above / below: rows of the DEM an output row depends on.
// upload == false: the DEM is on the device already (later scales of a multi-scale call).
template <class Compute>
int run_pipelined(HostRun& run, const float* dem, float* d_in, int ny, int nx, int above, int below, bool upload,
                  const std::vector<HostPlane>& outs, Compute&& compute) {
    Context& c = ctx();
    const int chunk = pipeline_chunk_rows(ny, nx);
    const std::vector<int> cut = chunk >= ny ? std::vector<int>{0, ny} : pipeline_cuts(ny, chunk);
    const int nchunks = (int)cut.size() - 1;
    const size_t row_bytes = (size_t)nx * sizeof(float);
    (void)above;
    t_host_chunks = std::max(t_host_chunks, nchunks < 3 ? 1 : nchunks);  // (a multi-scale call: the most any of its scales ran in)
    if (nchunks < 3) {
        if (upload) TOPO_HIP(hipMemcpyAsync(d_in, dem, (size_t)ny * row_bytes, hipMemcpyHostToDevice, c.compute));
        const int rc = compute(ny, 0, ny);
        if (rc != TOPO_AMD_OK && rc != TOPO_AMD_EEMPTY) return rc;
        run.ready();
        for (const HostPlane& o : outs) TOPO_TRY(download(o.host, o.dev, (size_t)ny * row_bytes));
        TOPO_HIP(hipStreamSynchronize(c.compute));
        return rc;
    }
    if (!c.up) {
        TOPO_HIP(hipStreamCreateWithFlags(&c.up, hipStreamNonBlocking));
        TOPO_HIP(hipStreamCreateWithFlags(&c.down, hipStreamNonBlocking));
    }
    while ((int)c.pipe_events.size() < 2 * nchunks) {
        hipEvent_t e = nullptr;
        TOPO_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c.pipe_events.push_back(e);
    }
    hipEvent_t* up_done = c.pipe_events.data();
    hipEvent_t* computed = c.pipe_events.data() + nchunks;
    // Page-locked arrays (topo_amd_host_alloc, hipHostRegister): every copy is asynchronous, and the calling thread issues the
    // downloads itself, behind each chunk's kernels.  Pageable arrays: copies block their caller, so the downloads go to a
    // second thread (tools/ubench/pipe_paths.hip: 23.4 / 23.9 ms for 2 x 1 GiB against 38 - 39 ms one after the other).
    // TOPO_AMD_HOST_DOWNLOADS=thread / inline forces one or the other.
    const int forced_mode = [] {
        const char* e = std::getenv("TOPO_AMD_HOST_DOWNLOADS");
        return e && *e == 't' ? 1 : (e && *e == 'i' ? 2 : 0);
    }();
    auto page_locked = [](const void* p) {
        if (!p) return true;
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return attr.type == hipMemoryTypeHost;
    };
    bool all_locked = !upload || page_locked(dem);
    for (const HostPlane& o : outs) all_locked = all_locked && page_locked(o.host);
    const bool inline_downloads = forced_mode == 2 || (forced_mode == 0 && all_locked);
    std::mutex mu;
    std::condition_variable cv;
    int ready_chunks = 0;  // chunks whose kernels are enqueued and whose event is recorded
    bool failed = false;
    int down_rc = TOPO_AMD_OK;
    std::string down_error;
    const int device = c.device;
    if (inline_downloads) run.ready();
    std::thread downloader([&] {
        if (inline_downloads) return;
        (void)hipSetDevice(device);
        run.ready();  // the result arrays have their pages
        for (int j = 0; j < nchunks; ++j) {
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return ready_chunks > j || failed; });
                if (failed) return;
            }
            const int r0 = cut[j], r1 = cut[j + 1];
            hipError_t e = hipStreamWaitEvent(c.down, computed[j], 0);
            for (size_t k = 0; k < outs.size() && e == hipSuccess; ++k)
                if (outs[k].host)
                    e = hipMemcpyAsync(outs[k].host + (size_t)r0 * nx, outs[k].dev + (size_t)r0 * nx, (size_t)(r1 - r0) * row_bytes,
                                       hipMemcpyDeviceToHost, c.down);
            if (e != hipSuccess) {
                down_rc = TOPO_AMD_EHIP;
                down_error = std::string("download of a row chunk failed: ") + hipGetErrorString(e);
                return;
            }
        }
        if (hipStreamSynchronize(c.down) != hipSuccess) {
            down_rc = TOPO_AMD_EHIP;
            down_error = "hipStreamSynchronize(download stream) failed";
        }
    });
    int rc = TOPO_AMD_OK, next = 0;
    bool empty = false;
    auto fail = [&](int code) {
        rc = code;
        std::lock_guard<std::mutex> lock(mu);
        failed = true;
    };
    for (int k = 0; k < nchunks && rc == TOPO_AMD_OK; ++k) {
        const int u0 = cut[k], u1 = cut[k + 1];
        if (upload) {
            hipError_t e = hipMemcpyAsync(d_in + (size_t)u0 * nx, dem + (size_t)u0 * nx, (size_t)(u1 - u0) * row_bytes,
                                          hipMemcpyHostToDevice, c.up);
            if (e == hipSuccess) e = hipEventRecord(up_done[k], c.up);
            if (e != hipSuccess) {
                set_error("upload of a row chunk failed: %s", hipGetErrorString(e));
                fail(TOPO_AMD_EHIP);
                break;
            }
        }
        const int uploaded = upload ? u1 : ny;
        while (next < nchunks && std::min(ny, cut[next + 1] + below) <= uploaded) {
            const int r0 = cut[next], r1 = cut[next + 1];
            if (upload && hipStreamWaitEvent(c.compute, up_done[k], 0) != hipSuccess) {
                set_error("hipStreamWaitEvent(compute, upload) failed");
                fail(TOPO_AMD_EHIP);
                break;
            }
            const int r = compute(uploaded, r0, r1 - r0);
            if (r != TOPO_AMD_OK && r != TOPO_AMD_EEMPTY) {
                fail(r);
                break;
            }
            if (r == TOPO_AMD_EEMPTY) empty = true;  // (Sx: every plane was written; reported at the end)
            if (hipEventRecord(computed[next], c.compute) != hipSuccess) {
                set_error("hipEventRecord(computed chunk) failed");
                fail(TOPO_AMD_EHIP);
                break;
            }
            if (inline_downloads) {
                hipError_t e = hipStreamWaitEvent(c.down, computed[next], 0);
                for (size_t q = 0; q < outs.size() && e == hipSuccess; ++q)
                    if (outs[q].host)
                        e = hipMemcpyAsync(outs[q].host + (size_t)r0 * nx, outs[q].dev + (size_t)r0 * nx, (size_t)(r1 - r0) * row_bytes,
                                           hipMemcpyDeviceToHost, c.down);
                if (e != hipSuccess) {
                    set_error("download of a row chunk failed: %s", hipGetErrorString(e));
                    fail(TOPO_AMD_EHIP);
                    break;
                }
            }
            {
                std::lock_guard<std::mutex> lock(mu);
                ready_chunks = ++next;
            }
            cv.notify_all();
        }
    }
    cv.notify_all();
    downloader.join();
    if (inline_downloads && hipStreamSynchronize(c.down) != hipSuccess && rc == TOPO_AMD_OK) {
        set_error("hipStreamSynchronize(download stream) failed");
        rc = TOPO_AMD_EHIP;
    }
    (void)hipStreamSynchronize(c.compute);
    if (upload) (void)hipStreamSynchronize(c.up);
    if (rc != TOPO_AMD_OK) return rc;
    if (down_rc != TOPO_AMD_OK) {
        set_error("%s", down_error.c_str());
        return down_rc;
    }
    return empty ? TOPO_AMD_EEMPTY : TOPO_AMD_OK;
}

}  // namespace

void note_valley_route(int route) { t_valley_route = route; }

}  // namespace topo

using namespace topo;

namespace {
int check_gate_errors();
uint32_t g_giveups_reported = 0;  // value of the give-up counter at the last topo_amd_gate_giveups
uint32_t g_giveups_probed = 0;    //                              ... at the last look of the mode logic
int g_probe_calls = 0;            // careful calls whose completion nobody has looked at yet (the probe event stands behind the last of them)
}

extern "C" {

const char* topo_amd_version(void) { return "topo_amd 0.1.0 (gfx950)"; }
const char* topo_amd_last_error(void) { return g_error.c_str(); }

int topo_amd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int topo_amd_init(int device) {
    TOPO_ENTER();
    Context& c = ctx();
    if (c.ready && c.device == device) return TOPO_AMD_OK;
    if (c.ready) {
        set_error("topo_amd_init: already bound to device %d (one process drives one GPU)", c.device);
        return TOPO_AMD_EINVAL;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        set_error("topo_amd_init: no HIP device visible");
        return TOPO_AMD_ENODEV;
    }
    TOPO_REQUIRE(device >= 0 && device < n, "topo_amd_init: device %d not in [0, %d)", device, n);
    TOPO_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    TOPO_HIP(hipGetDeviceProperties(&prop, device));
    c.num_cu = prop.multiProcessorCount;
    // TOPO_AMD_CU_LIMIT=<n>: persistent kernels size their grids for at most n compute units (to
    // leave room for other work on the GPU; the results do not depend on it)
    if (const char* lim = std::getenv("TOPO_AMD_CU_LIMIT")) {
        const int n = std::atoi(lim);
        if (n >= 8 && n < c.num_cu) c.num_cu = n;
    }
    TOPO_HIP(hipStreamCreateWithFlags(&c.compute, hipStreamNonBlocking));
    {
        // the ghost-row exchange goes on a high-priority stream: RCCL's few workgroups are dispatched ahead of the
        // pending workgroups of a grid kernel on the compute stream (which otherwise refill every slot that frees up
        // and leave the exchange for the end of the launch: Sx, kernel trace in profiles/r04_shard_fused.txt)
        int least = 0, greatest = 0;
        TOPO_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        TOPO_HIP(hipStreamCreateWithPriority(&c.comm, hipStreamNonBlocking, greatest));
    }
    TOPO_HIP(hipEventCreateWithFlags(&c.halo_done, hipEventDisableTiming));
    TOPO_HIP(hipEventCreateWithFlags(&c.input_ready, hipEventDisableTiming));
    TOPO_HIP(hipEventCreate(&c.t0));
    TOPO_HIP(hipEventCreate(&c.t1));
    // the ghost-row gate (common.hpp): a device word the communication stream stamps with the exchange epoch, and a
    // pinned host word in which blocks that gave up waiting count themselves
    TOPO_HIP(hipMalloc((void**)&c.gate_word, kGateBytes));
    TOPO_HIP(hipMemset(c.gate_word, 0, kGateBytes));
    TOPO_HIP(hipHostMalloc((void**)&c.gate_timeouts, 64, hipHostMallocMapped));
    g_giveups_reported = g_giveups_probed = 0;
    g_probe_calls = 0;
    c.gate_timeouts[0] = 0;  // blocks that gave up at a closed gate (careful mode)
    c.gate_timeouts[4] = 0;  // blocks whose wait ran out (lean mode): an error
    c.gate_epoch = 0;
    c.gate_mode = 0;
    c.gate_clean_calls = 0;
    c.gate_probe_pending = false;
    TOPO_HIP(hipEventCreateWithFlags(&c.gate_probe, hipEventDisableTiming));
    c.device = device;
    c.ready = true;
    return TOPO_AMD_OK;
}

// Per-launch timing without a host synchronise per launch: numbered HIP events on the compute stream.
namespace {
constexpr int kMarks = 512;
hipEvent_t g_marks[kMarks] = {};
}  // namespace

int topo_amd_shutdown(void) {
    TOPO_ENTER();
    Context& c = ctx();
    if (!c.ready) return TOPO_AMD_OK;
    (void)hipDeviceSynchronize();
    if (g_comm.comm) {
        (void)ncclCommDestroy(g_comm.comm);
        g_comm = Comm();
    }
    {
        std::lock_guard<std::mutex> lock(g_memo_mu);
        g_declared.clear();
    }
    valley_fft_release();  // FFT plans hold the stream that goes away below
    for (int i = 0; i < 12; ++i)
        if (c.ws[i]) (void)hipFree(c.ws[i]);
    for (int i = 0; i < 6; ++i) {
        if (c.tab[i]) (void)hipFree(c.tab[i]);
        if (g_slots[i].pinned) (void)hipHostFree(g_slots[i].pinned);
        if (g_slots[i].copied) (void)hipEventDestroy(g_slots[i].copied);
        g_slots[i] = TableSlot();
    }
    for (auto& e : g_marks) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    if (g_memo_words) {
        (void)hipHostFree(g_memo_words);
        g_memo_words = nullptr;
        for (auto& m : g_memo) m = DemMemo();
    }
    if (c.gate_word) (void)hipFree(c.gate_word);
    if (c.gate_timeouts) (void)hipHostFree(c.gate_timeouts);
    if (c.gate_probe) (void)hipEventDestroy(c.gate_probe);
    (void)hipEventDestroy(c.halo_done);
    (void)hipEventDestroy(c.input_ready);
    (void)hipEventDestroy(c.t0);
    (void)hipEventDestroy(c.t1);
    (void)hipStreamDestroy(c.compute);
    (void)hipStreamDestroy(c.comm);
    for (void* q : c.host_planes)
        if (q) (void)hipFree(q);
    if (c.up) {
        (void)hipStreamDestroy(c.up);
        (void)hipStreamDestroy(c.down);
    }
    for (hipEvent_t e : c.pipe_events) (void)hipEventDestroy(e);
    if (c.aux) {
        (void)hipStreamDestroy(c.aux);
        for (auto& e : c.aux_ready) (void)hipEventDestroy(e);
        (void)hipEventDestroy(c.aux_done);
    }
    c = Context();
    return TOPO_AMD_OK;
}

int topo_amd_device_name(char* buf, int buflen) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    hipDeviceProp_t prop;
    TOPO_HIP(hipGetDeviceProperties(&prop, ctx().device));
    // (the runtime of this image reports an empty marketing name for the MI355X)
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name[0] ? prop.name : "AMD Instinct", prop.gcnArchName, prop.multiProcessorCount);
    return TOPO_AMD_OK;
}

int topo_amd_cu_count(void) {
    TOPO_ENTER();
    if (require_ready() != TOPO_AMD_OK) return TOPO_AMD_ENODEV;
    return ctx().num_cu;
}

int topo_amd_malloc(void** dptr, size_t bytes) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(dptr != nullptr, "topo_amd_malloc: NULL result pointer");
    TOPO_HIP(hipMalloc(dptr, bytes ? bytes : 4));
    dem_memo_forget(*dptr, bytes ? bytes : 4);
    return TOPO_AMD_OK;
}

int topo_amd_free(void* dptr) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    if (dptr) {
        TOPO_HIP(hipStreamSynchronize(ctx().compute));
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)dptr) == hipSuccess) dem_memo_forget(base, size);
        else (void)hipGetLastError();
        dem_memo_forget(dptr, 0);
        TOPO_HIP(hipFree(dptr));
    }
    return TOPO_AMD_OK;
}

// Page-locked host memory for the arrays handed to the host-buffer entry points (topo_amd_*_f32): the copies then run
// at the link's rate without the driver staging them (bench.py, end_to_end: pinned against pageable).
int topo_amd_host_alloc(void** hptr, size_t bytes) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(hptr != nullptr && bytes > 0, "host_alloc: bad arguments");
    TOPO_HIP(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return TOPO_AMD_OK;
}
int topo_amd_host_free(void* hptr) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    if (hptr) TOPO_HIP(hipHostFree(hptr));
    return TOPO_AMD_OK;
}

int topo_amd_release_host_planes(void) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    Context& c = ctx();
    TOPO_HIP(hipStreamSynchronize(c.compute));
    for (size_t k = 0; k < c.host_planes.size(); ++k) {
        if (c.host_planes[k]) {
            dem_memo_forget(c.host_planes[k], c.host_plane_bytes[k]);
            TOPO_HIP(hipFree(c.host_planes[k]));
        }
    }
    c.host_planes.clear();
    c.host_plane_bytes.clear();
    return TOPO_AMD_OK;
}

int topo_amd_valley_route(int* route) {
    TOPO_REQUIRE(route != nullptr, "valley_route: NULL output");
    *route = t_valley_route;
    return TOPO_AMD_OK;
}

int topo_amd_host_chunks(int* chunks) {
    TOPO_REQUIRE(chunks != nullptr, "host_chunks: NULL output");
    *chunks = t_host_chunks;
    return TOPO_AMD_OK;
}

int topo_amd_memcpy_h2d(void* dst, const void* src, size_t bytes) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    dem_memo_forget(dst, bytes);
    TOPO_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx().compute));
    TOPO_HIP(hipStreamSynchronize(ctx().compute));
    return TOPO_AMD_OK;
}

int topo_amd_memcpy_d2h(void* dst, const void* src, size_t bytes) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    {
        // a destination fresh from the allocator has no pages yet: fault them in from several
        // threads (the kernels launched before this call are usually still running meanwhile)
        HostRun pages(false);
        pages.prefault(dst, bytes);
    }
    TOPO_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx().compute));
    TOPO_HIP(hipStreamSynchronize(ctx().compute));
    return check_gate_errors();
}

int topo_amd_memcpy_d2d(void* dst, const void* src, size_t bytes) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    dem_memo_forget(dst, bytes);
    TOPO_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx().compute));
    return TOPO_AMD_OK;
}

int topo_amd_memset(void* dst, int value, size_t bytes) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    dem_memo_forget(dst, bytes);
    TOPO_HIP(hipMemsetAsync(dst, value, bytes, ctx().compute));
    return TOPO_AMD_OK;
}

namespace topo {
namespace {
void gate_probe_poll();
}
}  // namespace topo

namespace {
// lean mode: a block's wait at the gate ran out (gate.hpp) - the seam rows of that call were computed from ghost rows
// that had not arrived.  Reported once, by the next call that synchronises; the library is careful from then on.
int check_gate_errors() {
    Context& c = ctx();
    gate_probe_poll();  // (a synchronising call: whatever careful calls were in flight are over)
    volatile uint32_t* err = c.gate_timeouts ? c.gate_timeouts + 4 : nullptr;
    if (err && *err != 0) {
        const unsigned n = *err;
        *err = 0;
        c.gate_mode = 2;
        set_error("ghost-row gate: %u blocks waited longer than TOPO_AMD_GATE_TIMEOUT_MS for the halo exchange; the seam "
                  "rows of the sharded calls since the last synchronisation are invalid (is a neighbour rank missing?).  "
                  "Later calls run with the clean-up launch again", n);
        return TOPO_AMD_ERCCL;
    }
    return TOPO_AMD_OK;
}
}  // namespace

// Blocks of sharded launches that gave up waiting at the ghost-row gate since the last call (gate.hpp): their seam
// tiles were done by the clean-up launch behind the exchange, i.e. the exchange was NOT hidden behind the interior
// rows for them.  0 in a healthy run (after the first calls, in which RCCL sets its connections up); a statistic
// for bench.py and the tests, never an error.
int topo_amd_gate_giveups(unsigned* count) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(count != nullptr, "gate_giveups: NULL output");
    const uint32_t now = *(volatile uint32_t*)ctx().gate_timeouts;
    *count = now - g_giveups_reported;  // (both reset by topo_amd_init)
    g_giveups_reported = now;
    return TOPO_AMD_OK;
}

int topo_amd_sync(void) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_HIP(hipStreamSynchronize(ctx().compute));
    TOPO_HIP(hipStreamSynchronize(ctx().comm));
    return check_gate_errors();
}

int topo_amd_timer_start(void) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_HIP(hipEventRecord(ctx().t0, ctx().compute));
    return TOPO_AMD_OK;
}

int topo_amd_timer_stop(float* elapsed_ms) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_HIP(hipEventRecord(ctx().t1, ctx().compute));
    TOPO_HIP(hipEventSynchronize(ctx().t1));
    TOPO_HIP(hipEventElapsedTime(elapsed_ms, ctx().t0, ctx().t1));
    return check_gate_errors();
}


int topo_amd_mark(int index) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(index >= 0 && index < kMarks, "mark: index %d outside [0, %d)", index, kMarks);
    if (!g_marks[index]) TOPO_HIP(hipEventCreate(&g_marks[index]));
    TOPO_HIP(hipEventRecord(g_marks[index], ctx().compute));
    return TOPO_AMD_OK;
}

int topo_amd_mark_elapsed(int from, int to, float* elapsed_ms) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(from >= 0 && from < kMarks && to >= 0 && to < kMarks && g_marks[from] && g_marks[to] && elapsed_ms,
                 "mark_elapsed: marks %d and %d must have been recorded", from, to);
    TOPO_HIP(hipEventSynchronize(g_marks[to]));
    TOPO_HIP(hipEventElapsedTime(elapsed_ms, g_marks[from], g_marks[to]));
    return check_gate_errors();
}

int topo_amd_synth_dem_dev(float* out, int rows, int row0, int nx, uint32_t seed, int integer_valued) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(out && rows >= 1 && nx >= 1, "synth_dem: bad arguments");
    forget_plane(out, rows, nx);
    return launch_synth(out, rows, row0, nx, seed, integer_valued != 0);
}

// ---- the raster class (common.hpp) ----------------------------------------------------------------------------------
int topo_amd_dem_changed(const void* dptr, size_t bytes) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(dptr != nullptr, "dem_changed: NULL pointer");
    if (bytes == 0) {  // the whole allocation
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)dptr) == hipSuccess) dem_memo_forget(base, size);
        else (void)hipGetLastError();
    }
    dem_memo_forget(dptr, bytes);
    return TOPO_AMD_OK;
}

int topo_amd_raster_scan_dev(const float* in, int in_rows, int in_row0, int gny, int nx, int own_row0, int own_rows,
                             uint64_t counts[3], float range[2]) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(in && counts && range && gny >= 1 && nx >= 1 && in_rows >= 1 && in_row0 >= 0 && in_row0 + in_rows <= gny,
                 "raster_scan: bad block");
    TOPO_REQUIRE(own_rows >= 0 && own_row0 >= in_row0 && own_row0 + own_rows <= in_row0 + in_rows,
                 "raster_scan: rows [%d, %d) are not inside the block's rows [%d, %d)", own_row0, own_row0 + own_rows, in_row0,
                 in_row0 + in_rows);
    Block b{in, in_rows, in_row0, gny, nx, own_row0, own_rows};
    Scan s;
    s.lo = range[0];
    s.hi = range[1];
    TOPO_TRY(scan_block(b, own_row0, own_rows, &s));
    counts[0] += s.taken;
    counts[1] += s.large;
    counts[2] += s.frac;
    range[0] = s.lo;
    range[1] = s.hi;
    return TOPO_AMD_OK;
}

namespace {
int check_declared_rows(const float* block, int in_rows, int gny, int nx, const char* who) {
    TOPO_REQUIRE(block != nullptr && in_rows >= 1 && gny >= in_rows && nx >= 1,
                 "%s: the declaration is for device rows: block != NULL, 1 <= in_rows <= gny, nx >= 1 (got %p, %d, %d, %d)", who,
                 (const void*)block, in_rows, gny, nx);
    return TOPO_AMD_OK;
}
}  // namespace

int topo_amd_raster_class_set(const float* block, int in_rows, int gny, int nx, int large, float lo, float hi, float frac_share) {
    if (large < 0) {  // withdraw: what was declared for memory overlapping these rows (block == NULL: every declaration)
        std::lock_guard<std::mutex> lock(g_memo_mu);
        if (block == nullptr) g_declared.clear();
        else forget_declared((uintptr_t)block, (uintptr_t)block + (size_t)std::max(1, in_rows) * std::max(1, nx) * sizeof(float));
        return TOPO_AMD_OK;
    }
    TOPO_TRY(check_declared_rows(block, in_rows, gny, nx, "raster_class_set"));
    TOPO_REQUIRE(!(lo != lo) && !(hi != hi) && frac_share >= 0.0f && frac_share <= 1.0f, "raster_class_set: bad range or share");
    RasterClass c;
    c.large = large != 0;
    c.lo = lo;
    c.hi = hi;
    c.frac_share = frac_share;
    declare_class(block, in_rows, gny, nx, c);
    return TOPO_AMD_OK;
}

int topo_amd_raster_class_from_scan(const float* block, int in_rows, int gny, int nx, const uint64_t counts[3], const float range[2]) {
    TOPO_REQUIRE(counts && range, "raster_class_from_scan: NULL argument");
    TOPO_TRY(check_declared_rows(block, in_rows, gny, nx, "raster_class_from_scan"));
    Scan s;
    s.taken = counts[0];
    s.large = counts[1];
    s.frac = counts[2];
    s.lo = range[0];
    s.hi = range[1];
    declare_class(block, in_rows, gny, nx, class_of(s));
    return TOPO_AMD_OK;
}

int topo_amd_raster_class_get(const float* block, int gny, int nx, int* declared, int* large, float* lo, float* hi, float* frac_share) {
    TOPO_REQUIRE(block && declared && large && lo && hi && frac_share, "raster_class_get: NULL argument");
    RasterClass c;  // (nothing declared: what a partial block is then taken for)
    Block b{block, 1, 0, gny, nx, 0, 1};
    *declared = declared_class(b, &c) ? 1 : 0;
    *large = c.large ? 1 : 0;
    *lo = c.lo;
    *hi = c.hi;
    *frac_share = c.frac_share;
    return TOPO_AMD_OK;
}

// ---- geometry helpers ---------------------------------------------------------------------
int topo_amd_disc_tap_count(int size) {
    DiscRuns d;
    if (build_disc(size, &d) != TOPO_AMD_OK) return TOPO_AMD_EINVAL;
    return d.taps;
}

int topo_amd_disc_mask(int size, float* mask) {
    TOPO_REQUIRE(mask != nullptr, "disc_mask: NULL output");
    DiscRuns d;
    TOPO_TRY(build_disc(size, &d));
    const int c = (size - 1) / 2;
    for (int a = 0; a < size; ++a) {
        const int row = (c - a) - d.dj_min;
        for (int b = 0; b < size; ++b) {
            const int di = c - b;
            mask[a * size + b] = (di >= d.lo[row] && di <= d.hi[row]) ? 1.0f : 0.0f;
        }
    }
    return TOPO_AMD_OK;
}

int topo_amd_halo_rows(int descriptor, double p0, double p1, int* above, int* below) {
    TOPO_REQUIRE(above && below, "halo_rows: NULL output");
    switch (descriptor) {
        case TOPO_AMD_DESC_TPI:
        case TOPO_AMD_DESC_STD: {
            DiscRuns d;
            TOPO_TRY(build_disc((int)p0, &d));
            int R = p1 > 0.0 ? gaussian_radius(p1) : 0;
            if (R >= mfma_min_radius(false) && R < 16) R = 16;  // pre-smoothing on the matrix cores: see DESC_GAUSS
            *above = -d.dj_min + R;
            *below = d.dj_max + R;
            return TOPO_AMD_OK;
        }
        case TOPO_AMD_DESC_GAUSS: {
            // (radius mfma_min_radius .. 15: the matrix-core kernels want the accumulation-offset row of every
            // 32-row tile inside the block, 16 rows from the tile's first row: gauss.hip, mfma_rows_ok)
            const int R = gaussian_radius(p0);
            *above = *below = (R >= mfma_min_radius(false) && R < 16) ? 16 : R;
            return TOPO_AMD_OK;
        }
        case TOPO_AMD_DESC_GRADIENT:
            *above = *below = gradient_shard_halo(p0, p1);
            return TOPO_AMD_OK;
        case TOPO_AMD_DESC_SOBEL:
            *above = *below = 1;
            return TOPO_AMD_OK;
        case TOPO_AMD_DESC_SX:
            *above = p0 > 0 ? (int)p0 : 0;
            *below = p1 > 0 ? (int)p1 : 0;
            return TOPO_AMD_OK;
        case TOPO_AMD_DESC_VALLEY_RIDGE: {
            const int32_t kmax = (int32_t)p0;
            TOPO_REQUIRE(kmax >= 1, "halo_rows: kernel side %d", kmax);
            (void)valley_ridge_reach(&kmax, 1, above, below);
            return TOPO_AMD_OK;
        }
        default:
            set_error("halo_rows: unknown descriptor %d", descriptor);
            return TOPO_AMD_EINVAL;
    }
}

// ---- device row-block entry points ------------------------------------------------------------
int topo_amd_tpi_std_dev(const float* in, int in_rows, int in_row0, int gny, int nx, int size,
                         int out_row0, int out_rows, float* tpi_out, float* std_out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    Block b{in, in_rows, in_row0, gny, nx, out_row0, out_rows};
    ClassScope cls(b);
    forget_plane(tpi_out, out_rows, nx);
    forget_plane(std_out, out_rows, nx);
    return tpi_std_block(b, size, 0.0, tpi_out, std_out);
}

int topo_amd_tpi_multi_dev(const float* in, int in_rows, int in_row0, int gny, int nx, int n_sizes, const int32_t* sizes,
                           int out_row0, int out_rows, float* const* tpi_outs) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(n_sizes >= 1 && sizes && tpi_outs, "tpi_multi: no sizes");
    for (int k = 0; k < n_sizes; ++k) TOPO_REQUIRE(tpi_outs[k], "tpi_multi: NULL output plane %d", k);
    Block b{in, in_rows, in_row0, gny, nx, out_row0, out_rows};
    ClassScope cls(b);
    for (int k = 0; k < n_sizes; ++k) forget_plane(tpi_outs[k], out_rows, nx);
    // sizes that have a two-disc kernel (disc_pair.hip) go through it in pairs - the smallest with the next one up,
    // so that the shared ring is no larger than it has to be - the rest one by one
    std::vector<int> order(n_sizes), left;
    for (int k = 0; k < n_sizes; ++k) order[k] = k;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return sizes[x] < sizes[y]; });
    std::vector<char> done(n_sizes, 0);
    constexpr bool pairs_on = true;
    for (int i = 0; i + 1 < n_sizes && pairs_on; ++i) {
        const int a = order[i], c = order[i + 1];
        if (done[a] || done[c] || !tpi_pair_covers(sizes[a], sizes[c])) continue;
        DiscRuns da, dc;
        TOPO_TRY(build_disc(sizes[a], &da));
        TOPO_TRY(build_disc(sizes[c], &dc));
        TOPO_TRY(check_block(b, -dc.dj_min, dc.dj_max, "tpi_multi"));
        const int rc = launch_tpi_pair(b, sizes[a], tpi_outs[a], sizes[c], tpi_outs[c]);
        if (rc == TOPO_AMD_EUNSUP) break;  // planes the 16-byte row accesses cannot take: single launches handle them
        if (rc != TOPO_AMD_OK) return rc;
        done[a] = done[c] = 1;
        ++i;
    }
    for (int k = 0; k < n_sizes; ++k)
        if (!done[k]) TOPO_TRY(tpi_std_block(b, sizes[k], 0.0, tpi_outs[k], nullptr));
    return TOPO_AMD_OK;
}

int topo_amd_gaussian_dev(const float* in, int in_rows, int in_row0, int gny, int nx,
                          double sigma_y, double sigma_x, int out_row0, int out_rows, float* out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(out != nullptr, "gaussian: NULL output");
    TOPO_REQUIRE(sigma_y >= 0.0 && sigma_x >= 0.0, "gaussian: negative sigma");
    Block b{in, in_rows, in_row0, gny, nx, out_row0, out_rows};
    const int R = gaussian_radius(sigma_y);
    TOPO_TRY(check_block(b, R, R, "gaussian"));
    ClassScope cls(b);
    forget_plane(out, out_rows, nx);
    return launch_gaussian(b, sigma_y, sigma_x, out);
}

int topo_amd_sobel_dev(const float* in, int in_rows, int in_row0, int gny, int nx, int out_row0,
                       int out_rows, float* dx_out, float* dy_out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    Block b{in, in_rows, in_row0, gny, nx, out_row0, out_rows};
    TOPO_TRY(check_block(b, 1, 1, "sobel"));
    forget_plane(dx_out, out_rows, nx);
    forget_plane(dy_out, out_rows, nx);
    return launch_sobel(b, dx_out, dy_out);
}

int topo_amd_gradient_dev(const float* in, int in_rows, int in_row0, int gny, int nx, double sigma,
                          double sig_ratio, int res_mode, const void* res_x, const void* res_y,
                          int out_row0, int out_rows, float* dx_out, float* dy_out,
                          float* slope_out, float* aspect_out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    Block b{in, in_rows, in_row0, gny, nx, out_row0, out_rows};
    const int h = gradient_halo(sigma, sig_ratio);
    TOPO_TRY(check_block(b, h, h, "gradient"));
    ClassScope cls(b);
    for (float* o : {dx_out, dy_out, slope_out, aspect_out}) forget_plane(o, out_rows, nx);
    return launch_gradient(b, sigma, sig_ratio, res_mode, res_x, res_y, dx_out, dy_out, slope_out,
                           aspect_out);
}

int topo_amd_sx_dev(const float* in, int in_rows, int in_row0, int gny, int nx, const int32_t* dj,
                    const int32_t* di, const double* dist, int n_off, int window, double height,
                    int out_row0, int out_rows, float* out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(out && n_off >= 0 && (n_off == 0 || (dj && di && dist)), "sx: NULL argument");
    int up = 0, down = 0;
    for (int n = 0; n < n_off; ++n) {
        if (std::isnan(dist[n])) continue;
        up = std::max(up, -dj[n]);
        down = std::max(down, dj[n]);
    }
    Block b{in, in_rows, in_row0, gny, nx, out_row0, out_rows};
    // rows of the zero frame need no neighbours, interior rows never reach outside the DEM
    TOPO_TRY(check_block(b, up, down, "sx"));
    forget_plane(out, out_rows, nx);
    return launch_sx(b, dj, di, dist, n_off, window, height, out);
}

namespace {
// rows above / below an output row that the ray pixels of all sectors reach
void sx_multi_reach(int n_az, const int32_t* first, const int32_t* dj, const double* dist, int* up, int* down) {
    *up = *down = 0;
    for (int n = first[0]; n < first[n_az]; ++n) {
        if (std::isnan(dist[n])) continue;
        *up = std::max(*up, -dj[n]);
        *down = std::max(*down, dj[n]);
    }
}
}  // namespace

int topo_amd_sx_multi_dev(const float* in, int in_rows, int in_row0, int gny, int nx, int n_az,
                          const int32_t* first, const int32_t* dj, const int32_t* di, const double* dist,
                          const int32_t* window, double height, int out_row0, int out_rows,
                          float* const* outs) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(n_az >= 1 && first && dj && di && dist && window && outs, "sx_multi: NULL argument");
    for (int k = 0; k < n_az; ++k) TOPO_REQUIRE(outs[k], "sx_multi: NULL output plane %d", k);
    int up = 0, down = 0;
    sx_multi_reach(n_az, first, dj, dist, &up, &down);
    Block b{in, in_rows, in_row0, gny, nx, out_row0, out_rows};
    TOPO_TRY(check_block(b, up, down, "sx_multi"));
    for (int k = 0; k < n_az; ++k) forget_plane(outs[k], out_rows, nx);
    return launch_sx_multi(b, n_az, first, dj, di, dist, window, height, outs);
}

int topo_amd_valley_ridge_dev(const float* in, int in_rows, int in_row0, int gny, int nx, const float* taps,
                              const int32_t* ksize, const float* angles, int n_angles, int n_planes, double mean,
                              double stdev, int out_row0, int out_rows, float* norm_out, float* dir_out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(taps && ksize && angles && norm_out && dir_out && n_angles >= 1, "valley_ridge: NULL argument");
    int up = 0, down = 0;
    (void)valley_ridge_reach(ksize, n_angles, &up, &down);
    Block b{in, in_rows, in_row0, gny, nx, out_row0, out_rows};
    TOPO_TRY(check_block(b, up, down, "valley_ridge"));
    forget_plane(norm_out, out_rows, nx);
    forget_plane(dir_out, out_rows, nx);
    return launch_valley_ridge(b, taps, ksize, angles, n_angles, n_planes, mean, stdev, norm_out, dir_out);
}

int topo_amd_mean_std_dev(const float* in, size_t count, double* mean, double* stdev) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(in && mean && stdev && count >= 1, "mean_std: bad arguments");
    return launch_mean_std(in, count, mean, stdev);
}

// ---- host-buffer entry points ----------------------------------------------------------------
int topo_amd_tpi_std_f32(const float* dem, int ny, int nx, int size, double sigma, float* tpi_out,
                         float* std_out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(dem && ny >= 1 && nx >= 1, "tpi_std: bad DEM");
    TOPO_REQUIRE(tpi_out || std_out, "tpi_std: both outputs are NULL");
    const size_t bytes = (size_t)ny * nx * sizeof(float);
    ClassScope cls(scan_host(dem, ny, nx));  // the raster class from the caller's array: no launch, no synchronisation
    int above = 0, below = 0;
    TOPO_TRY(topo_amd_halo_rows(TOPO_AMD_DESC_TPI, (double)size, sigma, &above, &below));
    HostRun run;
    void *d_in = nullptr, *d_tpi = nullptr, *d_std = nullptr;
    TOPO_TRY(run.alloc(&d_in, bytes));
    if (tpi_out) TOPO_TRY(run.alloc(&d_tpi, bytes));
    if (std_out) TOPO_TRY(run.alloc(&d_std, bytes));
    run.prefault(tpi_out, bytes);
    run.prefault(std_out, bytes);
    return run_pipelined(run, dem, (float*)d_in, ny, nx, above, below, true, {{tpi_out, (float*)d_tpi}, {std_out, (float*)d_std}},
                         [&](int view_rows, int r0, int rows) {
                             Block b{(const float*)d_in, view_rows, 0, ny, nx, r0, rows};
                             return tpi_std_block(b, size, sigma, shift((float*)d_tpi, r0, nx), shift((float*)d_std, r0, nx));
                         });
}

int topo_amd_tpi_std_multi_f32(const float* dem, int ny, int nx, int n_scales, const int32_t* sizes,
                               const double* sigmas, float* const* tpi_outs, float* const* std_outs) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(dem && ny >= 1 && nx >= 1, "tpi_std_multi: bad DEM");
    TOPO_REQUIRE(n_scales >= 1 && sizes, "tpi_std_multi: no scales");
    TOPO_REQUIRE(tpi_outs || std_outs, "tpi_std_multi: both output lists are NULL");
    bool any_tpi = false, any_std = false;
    for (int k = 0; k < n_scales; ++k) {
        const bool t = tpi_outs && tpi_outs[k], s = std_outs && std_outs[k];
        TOPO_REQUIRE(t || s, "tpi_std_multi: scale %d has no output plane", k);
        any_tpi |= t;
        any_std |= s;
    }
    const size_t bytes = (size_t)ny * nx * sizeof(float);
    ClassScope cls(scan_host(dem, ny, nx));
    HostRun run;
    void *d_in = nullptr, *d_tpi = nullptr, *d_std = nullptr;
    TOPO_TRY(run.alloc(&d_in, bytes));
    if (any_tpi) TOPO_TRY(run.alloc(&d_tpi, bytes));
    if (any_std) TOPO_TRY(run.alloc(&d_std, bytes));
    for (int k = 0; k < n_scales; ++k) {
        if (tpi_outs && tpi_outs[k]) run.prefault(tpi_outs[k], bytes);
        if (std_outs && std_outs[k]) run.prefault(std_outs[k], bytes);
    }
    // scale 0 rides on the upload; every scale overlaps its kernels with its own downloads (the two device planes are
    // reused, so a scale starts when the one before it is down)
    for (int k = 0; k < n_scales; ++k) {
        float* t = tpi_outs && tpi_outs[k] ? (float*)d_tpi : nullptr;
        float* sd = std_outs && std_outs[k] ? (float*)d_std : nullptr;
        int above = 0, below = 0;
        TOPO_TRY(topo_amd_halo_rows(TOPO_AMD_DESC_TPI, (double)sizes[k], sigmas ? sigmas[k] : 0.0, &above, &below));
        TOPO_TRY(run_pipelined(run, dem, (float*)d_in, ny, nx, above, below, k == 0,
                               {{t ? tpi_outs[k] : nullptr, t}, {sd ? std_outs[k] : nullptr, sd}}, [&](int view_rows, int r0, int rows) {
                                   Block b{(const float*)d_in, view_rows, 0, ny, nx, r0, rows};
                                   return tpi_std_block(b, sizes[k], sigmas ? sigmas[k] : 0.0, shift(t, r0, nx), shift(sd, r0, nx));
                               }));
    }
    return TOPO_AMD_OK;
}

int topo_amd_tpi_f32(const float* dem, int ny, int nx, int size, double sigma, float* out) {
    TOPO_ENTER();
    TOPO_REQUIRE(out != nullptr, "tpi: NULL output");
    return topo_amd_tpi_std_f32(dem, ny, nx, size, sigma, out, nullptr);
}

int topo_amd_std_f32(const float* dem, int ny, int nx, int size, double sigma, float* out) {
    TOPO_ENTER();
    TOPO_REQUIRE(out != nullptr, "std: NULL output");
    return topo_amd_tpi_std_f32(dem, ny, nx, size, sigma, nullptr, out);
}

int topo_amd_gauss_f32(const float* dem, int ny, int nx, double sigma_y, double sigma_x, float* out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(dem && out && ny >= 1 && nx >= 1, "gauss: bad arguments");
    TOPO_REQUIRE(sigma_y >= 0.0 && sigma_x >= 0.0, "gaussian: negative sigma");
    const size_t bytes = (size_t)ny * nx * sizeof(float);
    ClassScope cls(scan_host(dem, ny, nx));
    int above = 0, below = 0;
    TOPO_TRY(topo_amd_halo_rows(TOPO_AMD_DESC_GAUSS, sigma_y, 0.0, &above, &below));
    HostRun run;
    void *d_in = nullptr, *d_out = nullptr;
    TOPO_TRY(run.alloc(&d_in, bytes));
    TOPO_TRY(run.alloc(&d_out, bytes));
    run.prefault(out, bytes);
    return run_pipelined(run, dem, (float*)d_in, ny, nx, above, below, true, {{out, (float*)d_out}}, [&](int view_rows, int r0, int rows) {
        return topo_amd_gaussian_dev((const float*)d_in, view_rows, 0, ny, nx, sigma_y, sigma_x, r0, rows, shift((float*)d_out, r0, nx));
    });
}

int topo_amd_sobel_f32(const float* dem, int ny, int nx, float* dx_out, float* dy_out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(dem && dx_out && dy_out && ny >= 1 && nx >= 1, "sobel: bad arguments");
    const size_t bytes = (size_t)ny * nx * sizeof(float);
    HostRun run;
    void *d_in = nullptr, *d_dx = nullptr, *d_dy = nullptr;
    TOPO_TRY(run.alloc(&d_in, bytes));
    TOPO_TRY(run.alloc(&d_dx, bytes));
    TOPO_TRY(run.alloc(&d_dy, bytes));
    run.prefault(dx_out, bytes);
    run.prefault(dy_out, bytes);
    return run_pipelined(run, dem, (float*)d_in, ny, nx, 1, 1, true, {{dx_out, (float*)d_dx}, {dy_out, (float*)d_dy}},
                         [&](int view_rows, int r0, int rows) {
                             return topo_amd_sobel_dev((const float*)d_in, view_rows, 0, ny, nx, r0, rows, shift((float*)d_dx, r0, nx),
                                                       shift((float*)d_dy, r0, nx));
                         });
}

int topo_amd_gradient_f32(const float* dem, int ny, int nx, double sigma, double sig_ratio,
                          int res_mode, const void* res_x, const void* res_y, float* dx_out,
                          float* dy_out, float* slope_out, float* aspect_out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(dem && ny >= 1 && nx >= 1, "gradient: bad DEM");
    const size_t bytes = (size_t)ny * nx * sizeof(float);
    ClassScope cls(scan_host(dem, ny, nx));
    HostRun run;
    void *d_in = nullptr, *d_o[4] = {nullptr, nullptr, nullptr, nullptr};
    float* host_out[4] = {dx_out, dy_out, slope_out, aspect_out};
    TOPO_TRY(run.alloc(&d_in, bytes));
    for (int k = 0; k < 4; ++k)
        if (host_out[k]) TOPO_TRY(run.alloc(&d_o[k], bytes));
    for (int k = 0; k < 4; ++k) run.prefault(host_out[k], bytes);
    const void *rx = res_x, *ry = res_y;
    if (res_mode == TOPO_AMD_RES_2D) {
        TOPO_REQUIRE(res_x && res_y, "gradient: resolution arrays are NULL");
        void *d_rx = nullptr, *d_ry = nullptr;
        TOPO_TRY(run.alloc(&d_rx, bytes));
        TOPO_TRY(run.alloc(&d_ry, bytes));
        TOPO_HIP(hipMemcpyAsync(d_rx, res_x, bytes, hipMemcpyHostToDevice, ctx().compute));
        TOPO_HIP(hipMemcpyAsync(d_ry, res_y, bytes, hipMemcpyHostToDevice, ctx().compute));
        rx = d_rx;
        ry = d_ry;
    }
    const int h = gradient_halo(sigma, sig_ratio == 0.0 ? 1.0 : sig_ratio);
    std::vector<HostPlane> outs;
    for (int k = 0; k < 4; ++k) outs.push_back({host_out[k], (float*)d_o[k]});
    return run_pipelined(run, dem, (float*)d_in, ny, nx, h, h, true, outs, [&](int view_rows, int r0, int rows) {
        const void *cx = rx, *cy = ry;
        if (res_mode == TOPO_AMD_RES_2D) {  // [out_rows x nx], aligned with the output rows
            cx = (const float*)rx + (size_t)r0 * nx;
            cy = (const float*)ry + (size_t)r0 * nx;
        }
        return topo_amd_gradient_dev((const float*)d_in, view_rows, 0, ny, nx, sigma, sig_ratio, res_mode, cx, cy, r0, rows,
                                     shift((float*)d_o[0], r0, nx), shift((float*)d_o[1], r0, nx), shift((float*)d_o[2], r0, nx),
                                     shift((float*)d_o[3], r0, nx));
    });
}

int topo_amd_sx_f32(const float* dem, int ny, int nx, const int32_t* dj, const int32_t* di,
                    const double* dist, int n_off, int window, double height, float* out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(dem && out && ny >= 1 && nx >= 1, "sx: bad arguments");
    // (no ray pixel at all: the tables may be NULL; the plane is zero-filled and the call answers TOPO_AMD_EEMPTY, like
    // a sector whose pixels are all NaN)
    TOPO_REQUIRE(n_off >= 0 && (n_off == 0 || (dj && di && dist)), "sx: NULL argument");
    const size_t bytes = (size_t)ny * nx * sizeof(float);
    int up = 0, down = 0;
    for (int n = 0; n < n_off; ++n) {
        if (std::isnan(dist[n])) continue;
        up = std::max(up, -dj[n]);
        down = std::max(down, dj[n]);
    }
    HostRun run;
    void *d_in = nullptr, *d_out = nullptr;
    TOPO_TRY(run.alloc(&d_in, bytes));
    TOPO_TRY(run.alloc(&d_out, bytes));
    run.prefault(out, bytes);
    return run_pipelined(run, dem, (float*)d_in, ny, nx, up, down, true, {{out, (float*)d_out}}, [&](int view_rows, int r0, int rows) {
        return topo_amd_sx_dev((const float*)d_in, view_rows, 0, ny, nx, dj, di, dist, n_off, window, height, r0, rows,
                               shift((float*)d_out, r0, nx));
    });
}

int topo_amd_sx_multi_f32(const float* dem, int ny, int nx, int n_az, const int32_t* first,
                          const int32_t* dj, const int32_t* di, const double* dist, const int32_t* window,
                          double height, float* const* outs) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(dem && outs && n_az >= 1 && ny >= 1 && nx >= 1, "sx_multi: bad arguments");
    TOPO_REQUIRE(first && dj && di && dist && window, "sx_multi: NULL argument");
    for (int k = 0; k < n_az; ++k) TOPO_REQUIRE(outs[k], "sx_multi: NULL output plane %d", k);
    const size_t bytes = (size_t)ny * nx * sizeof(float);
    HostRun run;
    void* d_in = nullptr;
    std::vector<float*> d_out(n_az, nullptr);
    TOPO_TRY(run.alloc(&d_in, bytes));
    for (int k = 0; k < n_az; ++k) TOPO_TRY(run.alloc((void**)&d_out[k], bytes));
    for (int k = 0; k < n_az; ++k) run.prefault(outs[k], bytes);
    int up = 0, down = 0;
    sx_multi_reach(n_az, first, dj, dist, &up, &down);
    std::vector<HostPlane> planes;
    for (int k = 0; k < n_az; ++k) planes.push_back({outs[k], d_out[k]});
    std::vector<float*> moved(n_az);
    return run_pipelined(run, dem, (float*)d_in, ny, nx, up, down, true, planes, [&](int view_rows, int r0, int rows) {
        for (int k = 0; k < n_az; ++k) moved[k] = shift(d_out[k], r0, nx);
        return topo_amd_sx_multi_dev((const float*)d_in, view_rows, 0, ny, nx, n_az, first, dj, di, dist, window, height, r0, rows,
                                     moved.data());  // (EEMPTY: every plane was written)
    });
}

int topo_amd_valley_ridge_f32(const float* dem, int ny, int nx, const float* taps, const int32_t* ksize,
                              const float* angles, int n_angles, int n_planes, double mean, double stdev,
                              float* norm_out, float* dir_out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(dem && ny >= 1 && nx >= 1 && norm_out && dir_out, "valley_ridge: bad arguments");
    const size_t bytes = (size_t)ny * nx * sizeof(float);
    HostRun run;
    void *d_in = nullptr, *d_norm = nullptr, *d_dir = nullptr;
    TOPO_TRY(run.alloc(&d_in, bytes));
    TOPO_TRY(run.alloc(&d_norm, bytes));
    TOPO_TRY(run.alloc(&d_dir, bytes));
    run.prefault(norm_out, bytes);
    run.prefault(dir_out, bytes);
    // (not pipelined: half a second of kernels per 20 ms of copies at full size, and the FFT route of the large kernels is
    // not cut-invariant)
    t_host_chunks = 1;
    TOPO_HIP(hipMemcpyAsync(d_in, dem, bytes, hipMemcpyHostToDevice, ctx().compute));
    TOPO_TRY(topo_amd_valley_ridge_dev((const float*)d_in, ny, 0, ny, nx, taps, ksize, angles, n_angles, n_planes,
                                       mean, stdev, 0, ny, (float*)d_norm, (float*)d_dir));
    run.ready();
    TOPO_TRY(download(norm_out, d_norm, bytes));
    TOPO_TRY(download(dir_out, d_dir, bytes));
    TOPO_HIP(hipStreamSynchronize(ctx().compute));
    return TOPO_AMD_OK;
}

// ---- RCCL row sharding ------------------------------------------------------------------------
namespace {
// CUs the persistent kernels of a sharded call leave to RCCL's send / receive kernel (TOPO_AMD_RESERVE_CUS, 0 ... 64;
// effective in steps of 8: one CU on every XCD, see march_grid)
int reserve_cus_setting() {
    static const int v = [] {
        const char* e = std::getenv("TOPO_AMD_RESERVE_CUS");
        return e && *e ? std::max(0, std::min(64, std::atoi(e))) : 8;
    }();
    return v;
}
// The ghost-row exchange is two point-to-point messages of a few MB over one xGMI link each (50 GB/s): it needs a
// few channels, not the 64 workgroups RCCL launches by default - which do not fit next to a persistent kernel
// that fills the chip, so that the "overlapped" exchange of rounds 1-3 really ran when the interior launch ended
// (kernel trace: profiles/r04_shard_fused.txt).  8 channels = 8 workgroups (256 threads, 37 KB of LDS, 124 VGPRs),
// one per XCD, each on the CU the shorter grid leaves free there; in loop-back they move the 2 x 4.3 MB of the TPI
// halo in 46 us (4 channels: 79 us, 2: 163 us).  Whatever the user has set stays.
void cap_rccl_channels() {
    setenv("NCCL_MAX_NCHANNELS", "8", 0);
    setenv("NCCL_MAX_P2P_NCHANNELS", "8", 0);
    setenv("NCCL_MIN_P2P_NCHANNELS", "1", 0);
}
}  // namespace

int topo_amd_comm_unique_id(char id[TOPO_AMD_UNIQUE_ID_BYTES]) {
    TOPO_ENTER();
    static_assert(sizeof(ncclUniqueId) <= TOPO_AMD_UNIQUE_ID_BYTES, "unique id size");
    cap_rccl_channels();
    ncclUniqueId uid;
    TOPO_NCCL(ncclGetUniqueId(&uid));
    std::memset(id, 0, TOPO_AMD_UNIQUE_ID_BYTES);
    std::memcpy(id, &uid, sizeof(uid));
    return TOPO_AMD_OK;
}

int topo_amd_comm_init(int rank, int nranks, const char id[TOPO_AMD_UNIQUE_ID_BYTES]) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: rank %d of %d", rank, nranks);
    TOPO_REQUIRE(g_comm.comm == nullptr, "comm_init: communicator already exists");
    cap_rccl_channels();
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    TOPO_NCCL(ncclCommInitRank(&g_comm.comm, nranks, uid, rank));
    g_comm.rank = rank;
    g_comm.size = nranks;
    return TOPO_AMD_OK;
}

int topo_amd_comm_rank(void) { return g_comm.rank; }
int topo_amd_comm_size(void) { return g_comm.size; }

int topo_amd_comm_destroy(void) {
    TOPO_ENTER();
    if (g_comm.comm) {
        (void)hipDeviceSynchronize();
        TOPO_NCCL(ncclCommDestroy(g_comm.comm));
    }
    g_comm = Comm();
    return TOPO_AMD_OK;
}

namespace {
__global__ void stamp_kernel(uint32_t* word, uint32_t value) {
    __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// writes the current epoch into the gate word, in stream order on the communication stream: a stream memory
// operation where the runtime has one (no kernel, no CU), a one-thread kernel otherwise
int stamp_gate(Context& c) {
    static int mode = 0;
    if (mode == 0) {
        if (hipStreamWriteValue32(c.comm, c.gate_word, c.gate_epoch, 0) == hipSuccess) return TOPO_AMD_OK;
        (void)hipGetLastError();
        mode = 1;
    }
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, c.comm, c.gate_word, c.gate_epoch);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}
}  // namespace

int topo_amd_halo_exchange_start(float* block, int rows_local, int nx, int halo_above,
                                 int halo_below) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    Context& c = ctx();
    TOPO_REQUIRE(block && rows_local >= 1 && nx >= 1 && halo_above >= 0 && halo_below >= 0,
                 "halo_exchange: bad arguments");
    TOPO_REQUIRE(rows_local >= halo_above && rows_local >= halo_below,
                 "halo_exchange: %d local rows cannot feed ghost depths %d/%d (shard too thin)",
                 rows_local, halo_above, halo_below);
    // the local rows must be final before neighbours read them
    TOPO_HIP(hipEventRecord(c.input_ready, c.compute));
    TOPO_HIP(hipStreamWaitEvent(c.comm, c.input_ready, 0));
    float* ghost_top = block;
    float* local = block + (size_t)halo_above * nx;
    float* ghost_bot = local + (size_t)rows_local * nx;
    if (halo_loopback()) {
        // the rank is its own upper and lower neighbour: same calls, same counts, periodic wrap.  Sends and
        // receives to one peer inside a group match in order.
        TOPO_NCCL(ncclGroupStart());
        if (halo_above > 0) {
            TOPO_NCCL(ncclSend(local + (size_t)(rows_local - halo_above) * nx, (size_t)halo_above * nx, ncclFloat, 0,
                               g_comm.comm, c.comm));
            TOPO_NCCL(ncclRecv(ghost_top, (size_t)halo_above * nx, ncclFloat, 0, g_comm.comm, c.comm));
        }
        if (halo_below > 0) {
            TOPO_NCCL(ncclSend(local, (size_t)halo_below * nx, ncclFloat, 0, g_comm.comm, c.comm));
            TOPO_NCCL(ncclRecv(ghost_bot, (size_t)halo_below * nx, ncclFloat, 0, g_comm.comm, c.comm));
        }
        TOPO_NCCL(ncclGroupEnd());
    } else if (g_comm.size > 1) {
        TOPO_REQUIRE(g_comm.comm != nullptr, "halo_exchange: call topo_amd_comm_init first");
        const int up = g_comm.rank - 1, down = g_comm.rank + 1;
        TOPO_NCCL(ncclGroupStart());
        if (up >= 0) {
            // my first halo_below rows become the upper neighbour's bottom ghost rows
            if (halo_below > 0)
                TOPO_NCCL(ncclSend(local, (size_t)halo_below * nx, ncclFloat, up, g_comm.comm, c.comm));
            if (halo_above > 0)
                TOPO_NCCL(ncclRecv(ghost_top, (size_t)halo_above * nx, ncclFloat, up, g_comm.comm, c.comm));
        }
        if (down < g_comm.size) {
            if (halo_above > 0)
                TOPO_NCCL(ncclSend(local + (size_t)(rows_local - halo_above) * nx,
                                   (size_t)halo_above * nx, ncclFloat, down, g_comm.comm, c.comm));
            if (halo_below > 0)
                TOPO_NCCL(ncclRecv(ghost_bot, (size_t)halo_below * nx, ncclFloat, down, g_comm.comm, c.comm));
        }
        TOPO_NCCL(ncclGroupEnd());
    }
    // the gate of the fused seam parts: the epoch of this exchange, written behind the receives
    ++c.gate_epoch;
    TOPO_TRY(stamp_gate(c));
    TOPO_HIP(hipEventRecord(c.halo_done, c.comm));
    g_comm.halo_pending = true;
    return TOPO_AMD_OK;
}

int topo_amd_shard_layout(int halo_above, int halo_below) {
    TOPO_REQUIRE((halo_above >= 0 && halo_below >= 0) || (halo_above == -1 && halo_below == -1),
                 "shard_layout: ghost depths %d / %d (both >= 0, or -1 / -1 for 'as the descriptor needs')", halo_above,
                 halo_below);
    t_layout_above = halo_above;
    t_layout_below = halo_below;
    t_layout_set = true;
    g_layout_above.store(halo_above);
    g_layout_below.store(halo_below);
    return TOPO_AMD_OK;
}

int topo_amd_shard_layout_get(int* halo_above, int* halo_below) {
    TOPO_REQUIRE(halo_above && halo_below, "shard_layout_get: NULL output");
    *halo_above = layout_above();
    *halo_below = layout_below();
    return TOPO_AMD_OK;
}

// Collective: every rank scans the lattice points of the rows it owns, the scans are added up (three all-reduces of a few
// bytes, once per DEM), and the class of the WHOLE raster is declared for the calling thread's later topo_amd_shard_* calls -
// so every shard takes the kernels the single-GPU run of the raster takes.  `owned`: the first row the rank owns.
int topo_amd_shard_classify(const float* owned, int rows_local, int row0, int gny, int nx) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(owned && rows_local >= 1 && row0 >= 0 && row0 + rows_local <= gny && nx >= 1, "shard_classify: bad arguments");
    // (a shard that is not the whole raster needs its neighbours' samples: without a communicator - a ShardedDEM built
    // before topo_amd_comm_init - the class of the rows at hand would pass for the raster's.  Loop-back: the rank is its
    // own neighbour, its rows are all there is.)
    TOPO_REQUIRE(rows_local == gny || g_comm.size > 1 || halo_loopback(),
                 "shard_classify: rows [%d, %d) of a raster of %d rows, but no communicator exists (topo_amd_comm_init) - the other "
                 "shards' samples cannot be added", row0, row0 + rows_local, gny);
    Scan s;
    if (halo_loopback() && rows_local < gny) {
        // the rank is every rank: the raster is its rows stacked periodically, so they are scanned at every placement
        for (long place = (long)row0 - ((long)row0 + rows_local - 1) / rows_local * rows_local; place < gny; place += rows_local) {
            const int o0 = (int)std::max(0L, place), o1 = (int)std::min((long)gny, place + rows_local);
            if (o1 <= o0) continue;
            Block b{owned, rows_local, (int)place, gny, nx, o0, o1 - o0};
            TOPO_TRY(scan_block(b, o0, o1 - o0, &s));
        }
    } else {
        Block b{owned, rows_local, row0, gny, nx, row0, rows_local};
        TOPO_TRY(scan_block(b, row0, rows_local, &s));
    }
    if (g_comm.size > 1) {
        TOPO_REQUIRE(g_comm.comm != nullptr, "shard_classify: call topo_amd_comm_init first");
        Context& c = ctx();
        void* d = nullptr;
        TOPO_TRY(workspace(0, 64, &d));
        struct Wire {
            unsigned long long counts[3];
            float lo, hi;
        } w{{s.taken, s.large, s.frac}, s.lo, s.hi};
        TOPO_HIP(hipMemcpyAsync(d, &w, sizeof(w), hipMemcpyHostToDevice, c.compute));
        char* base = (char*)d;
        TOPO_NCCL(ncclAllReduce(base, base, 3, ncclUint64, ncclSum, g_comm.comm, c.compute));
        TOPO_NCCL(ncclAllReduce(base + 24, base + 24, 1, ncclFloat, ncclMin, g_comm.comm, c.compute));
        TOPO_NCCL(ncclAllReduce(base + 28, base + 28, 1, ncclFloat, ncclMax, g_comm.comm, c.compute));
        TOPO_HIP(hipMemcpyAsync(&w, d, sizeof(w), hipMemcpyDeviceToHost, c.compute));
        TOPO_HIP(hipStreamSynchronize(c.compute));
        s.taken = w.counts[0];
        s.large = w.counts[1];
        s.frac = w.counts[2];
        s.lo = w.lo;
        s.hi = w.hi;
    }
    declare_class(owned, rows_local, gny, nx, class_of(s));
    return TOPO_AMD_OK;
}

int topo_amd_halo_wait(void) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    if (g_comm.halo_pending) {
        TOPO_HIP(hipStreamWaitEvent(ctx().compute, ctx().halo_done, 0));
        g_comm.halo_pending = false;
    }
    return TOPO_AMD_OK;
}

}  // extern "C"

namespace topo {
namespace {

// Geometry of one rank's haloed block inside the global DEM.
struct Shard {
    Block whole;           // block including the ghost rows that exist
    Block owned;           // the owned rows alone: what a launch may read while the exchange is in flight
    int row0, rows_local;  // owned rows
    int interior0, interior1;  // owned rows whose stencil stays inside the owned rows
};

// The buffer is laid out with the declared ghost depth (topo_amd_shard_layout); a descriptor that needs
// `above` rows uses the last `above` of the declared ones.  Returns where its [above | local | below] view
// starts.
int shard_view(float** block, int above, int below, const char* who) {
    TOPO_REQUIRE(*block != nullptr, "%s: NULL block", who);
    if (layout_above() < 0) return TOPO_AMD_OK;
    TOPO_REQUIRE(above <= layout_above() && below <= layout_below(),
                 "%s needs %d / %d ghost rows but the shard buffers were declared with %d / %d (topo_amd_shard_layout)",
                 who, above, below, layout_above(), layout_below());
    return TOPO_AMD_OK;
}
size_t shard_view_offset(int above, int nx) {
    return layout_above() < 0 ? 0 : (size_t)(layout_above() - above) * nx;
}

Shard make_shard(float* block, int rows_local, int row0, int gny, int nx, int above, int below) {
    Shard s;
    const bool has_up = row0 > 0, has_down = row0 + rows_local < gny;
    // the buffer always reserves `above` ghost rows on top; at the global edge they are unused
    s.whole.in = block + (size_t)(has_up ? 0 : above) * nx;
    s.whole.in_row0 = has_up ? row0 - above : row0;
    s.whole.in_rows = rows_local + (has_up ? above : 0) + (has_down ? below : 0);
    s.whole.gny = gny;
    s.whole.nx = nx;
    s.owned = s.whole;
    s.owned.in = block + (size_t)above * nx;
    s.owned.in_row0 = row0;
    s.owned.in_rows = rows_local;
    s.row0 = row0;
    s.rows_local = rows_local;
    s.interior0 = has_up ? std::min(row0 + above, row0 + rows_local) : row0;
    s.interior1 = has_down ? std::max(row0 + rows_local - below, s.interior0) : row0 + rows_local;
    return s;
}

// Runs `fn(block, out_row0, out_rows)` for the interior first, then (after the ghost rows landed) for the
// seam strips: the exchange overlaps the interior compute.  The interior launch gets the OWNED rows as
// its block: the ghost rows are being written by the exchange meanwhile, and a tile-based kernel stages
// (and classifies) every row of its block that a tile touches, not only the rows its outputs need.
// The three-launch route of rounds 1-3: interior rows first (on the owned rows: the ghost rows are being written
// meanwhile, and a tile-based kernel stages - and classifies - every row of its block that a tile touches), then, behind
// the exchange's event, the two seam strips.  `started`: the exchange is already in flight.
template <class Fn>
int run_three(float* block, const Shard& s, int above, int below, Fn fn, bool started) {
    if (!started) TOPO_TRY(topo_amd_halo_exchange_start(block, s.rows_local, s.whole.nx, above, below));
    // persistent launches leave a round of CUs to RCCL's kernel (march_grid)
    ctx().reserve_cus = (g_comm.size > 1 || halo_loopback()) ? reserve_cus_setting() : 0;
    int rc = TOPO_AMD_OK;
    if (s.interior1 > s.interior0) rc = fn(s.owned, s.interior0, s.interior1 - s.interior0);
    ctx().reserve_cus = 0;
    if (rc != TOPO_AMD_OK) return rc;
    TOPO_TRY(topo_amd_halo_wait());
    if (s.interior0 > s.row0) TOPO_TRY(fn(s.whole, s.row0, s.interior0 - s.row0));
    const int end = s.row0 + s.rows_local;
    if (end > s.interior1) TOPO_TRY(fn(s.whole, s.interior1, end - s.interior1));
    return TOPO_AMD_OK;
}
template <class Fn>
int run_overlapped(float* block, const Shard& s, int above, int below, Fn fn) {
    return run_three(block, s, above, below, fn, false);
}

// ---- round 4: ONE launch per kernel of the descriptor for the whole shard -------------------------------------------
// Two forms, both behind the ghost-row gate (gate.hpp) that the communication stream opens when the ghost rows have
// landed, both with the block views, output rows and kernels of run_three, hence its bits:
//  * run_fused (marching kernels: TPI / STD): the seam strips travel with the interior launch as "parts"
//    (Context::seams; disc_wave_impl.hpp, make_parts): every persistent block does its share of the interior, finds the
//    gate open - the exchange ran next to the interior - and does its share of the seams;
//  * run_gated (kernels of independent tiles: Sx): one launch over all owned rows on the block with its ghost rows; the
//    tile rows that read ghost rows are dispatched last and wait at the gate (Context::ghost).
// No relaunch tail, no under-filled seam launches, a third of the launches.  A launcher that cannot do it says so
// (leaves the gate armed / answers TOPO_AMD_EUNSUP before launching) and the call takes run_three.
// TOPO_AMD_SHARD_FUSED=0: always run_three (A/B).
bool shard_fused_on() {
    static const bool on = [] {
        const char* e = std::getenv("TOPO_AMD_SHARD_FUSED");
        return !(e && *e == '0');
    }();
    return on;
}
// how long a block waits at a closed gate before it leaves its seam tiles to the clean-up launch (TOPO_AMD_GATE_WAIT_US)
uint32_t gate_limit_ticks() {
    static const uint32_t v = [] {
        const char* e = std::getenv("TOPO_AMD_GATE_WAIT_US");
        const double us = e && *e ? std::atof(e) : 100.0;
        return (uint32_t)std::min(4.0e9, std::max(0.0, us) * 100.0);  // s_memtime: 100 MHz
    }();
    return v;
}
// lean mode: how long a block waits before it reports an error (TOPO_AMD_GATE_TIMEOUT_MS)
uint32_t gate_timeout_ticks() {
    static const uint32_t v = [] {
        const char* e = std::getenv("TOPO_AMD_GATE_TIMEOUT_MS");
        const double ms = e && *e ? std::atof(e) : 5000.0;  // (ranks drift apart by milliseconds, a first allocation by tens)
        return (uint32_t)std::min(4.0e9, std::max(1.0, ms) * 1.0e5);
    }();
    return v;
}
void gate_probe_poll() {
    Context& c = ctx();
    if (c.gate_mode != 0 || !c.gate_probe_pending || hipEventQuery(c.gate_probe) != hipSuccess) return;
    // every careful call up to the probe is over: did all of their blocks find the gate open?
    c.gate_probe_pending = false;
    const uint32_t now = *(volatile uint32_t*)c.gate_timeouts;
    c.gate_clean_calls = now == g_giveups_probed ? c.gate_clean_calls + g_probe_calls : 0;
    g_giveups_probed = now;
    g_probe_calls = 0;
    if (c.gate_clean_calls >= 3) c.gate_mode = 1;
}
// The gate of the exchange just started, careful or lean (Context::gate_mode; TOPO_AMD_GATE_MODE=careful / lean pins it).
Gate make_gate(bool* lean) {
    Context& c = ctx();
    // TOPO_AMD_GATE_MODE: careful (the default: a block that finds the gate closed for long leaves its seam tiles to the
    // clean-up launch behind the exchange's event - right whatever the neighbour ranks do, whoever reads the outputs),
    // lean (no clean-up launch, no event wait: a neighbour later than TOPO_AMD_GATE_TIMEOUT_MS is an ERROR that only
    // the next synchronising library call reports - for applications that do call one), auto (careful until a
    // few calls found the gate open, then lean).
    static const int pinned = [] {
        const char* e = std::getenv("TOPO_AMD_GATE_MODE");
        return e && e[0] == 'l' ? 1 : e && e[0] == 'a' ? 0 : 2;
    }();
    if (pinned) c.gate_mode = c.gate_mode == 2 ? 2 : pinned;  // (an error always ends lean mode)
    gate_probe_poll();
    *lean = c.gate_mode == 1;
    return Gate{c.gate_word, c.gate_epoch, c.gate_timeouts, *lean ? gate_timeout_ticks() : gate_limit_ticks(),
                (uint8_t*)c.gate_word + 256, *lean ? c.gate_timeouts + 4 : nullptr};
}
// behind a call whose launcher took the gate
int finish_gated(bool lean, bool probe) {
    Context& c = ctx();
    // every later launch on the compute stream is ordered behind a kernel all of whose blocks passed the gate
    if (lean) {
        g_comm.halo_pending = false;  // (no event wait on the compute stream: nothing is left to wait for)
        return TOPO_AMD_OK;
    }
    if (c.gate_mode == 0 && probe) {
        TOPO_HIP(hipEventRecord(c.gate_probe, c.compute));
        c.gate_probe_pending = true;
        ++g_probe_calls;
    }
    return topo_amd_halo_wait();  // (the clean-up launch waited already: keeps topo_amd_halo_wait's contract)
}

// tile_rows: the tile height of the descriptor's first kernel (a hint, 0 = unknown).  The kernels' tile rows sit on global
// multiples of it, so an interior that starts or ends inside a tile makes that tile be staged twice - once by the
// interior part for its lower rows, once by the seam part for its upper ones: the interior is shrunk to whole tiles and
// the two edge tiles belong to the seam parts entirely (any interior inside [interior0, interior1) is valid: the seam
// parts see the ghost rows).  68 + 2 tile slots per strip become 66 + 2 at 67 px on a 4096-row shard.
template <class Fn>
int run_fused(float* block, const Shard& s0, int above, int below, int tile_rows, Fn fn) {
    Shard s = s0;
    const int end = s.row0 + s.rows_local;
    if (tile_rows > 0) {
        const int a = s.interior0 > s.row0 ? (s.interior0 + tile_rows - 1) / tile_rows * tile_rows : s.interior0;
        const int b = end > s.interior1 ? s.interior1 / tile_rows * tile_rows : s.interior1;
        if (b > a) {
            s.interior0 = a;
            s.interior1 = b;
        }
    }
    const bool top = s.interior0 > s.row0, bottom = end > s.interior1;
    if (!shard_fused_on() || s.interior1 <= s.interior0 || (!top && !bottom)) return run_three(block, s0, above, below, fn, false);
    Context& c = ctx();
    TOPO_TRY(topo_amd_halo_exchange_start(block, s.rows_local, s.whole.nx, above, below));
    const bool live = g_comm.size > 1 || halo_loopback();
    c.reserve_cus = live ? reserve_cus_setting() : 0;
    c.seams = Seams();
    auto add = [&](int r0, int rows) {
        Block b = s.whole;
        b.out_row0 = r0;
        b.out_rows = rows;
        c.seams.b[c.seams.n++] = b;
    };
    if (top) add(s.row0, s.interior0 - s.row0);
    if (bottom) add(s.interior1, end - s.interior1);
    bool lean = false;
    c.seams.gate = make_gate(&lean);
    c.seams.gate_armed = true;
    const int rc = fn(s.owned, s.interior0, s.interior1 - s.interior0);
    const bool taken = !c.seams.gate_armed;
    c.seams = Seams();
    c.reserve_cus = 0;
    if (rc != TOPO_AMD_OK) return rc;
    if (taken) return finish_gated(lean, true);
    TOPO_TRY(topo_amd_halo_wait());
    if (top) TOPO_TRY(fn(s.whole, s.row0, s.interior0 - s.row0));
    if (bottom) TOPO_TRY(fn(s.whole, s.interior1, end - s.interior1));
    return TOPO_AMD_OK;
}

// device_gate: the launcher's kernel waits at the gate itself (Sx) - in lean mode only: a grid kernel keeps every CU
// slot filled with its own pending workgroups, so RCCL's kernel gets to run when the grid drains, i.e. when the
// blocks of the ghost tile rows are already waiting; with the careful mode's short wait they would all give up.
// Otherwise (gradient) the launcher orders its row chunks around the exchange's event and needs no mode.
template <class Fn>
int run_gated(float* block, const Shard& s, int above, int below, bool device_gate, Fn fn) {
    const int end = s.row0 + s.rows_local;
    const bool top = s.interior0 > s.row0, bottom = end > s.interior1;
    if (!shard_fused_on() || (!top && !bottom)) return run_three(block, s, above, below, fn, false);
    Context& c = ctx();
    bool lean = false;
    const Gate gate = make_gate(&lean);  // (the epoch is patched in below, once the exchange has its number)
    if (device_gate && !lean) return run_three(block, s, above, below, fn, false);
    TOPO_TRY(topo_amd_halo_exchange_start(block, s.rows_local, s.whole.nx, above, below));
    c.ghost = GhostGate();
    c.ghost.gate = gate;
    c.ghost.gate.epoch = c.gate_epoch;
    c.ghost.ghost_lo = s.whole.in_row0 < s.row0 ? s.row0 : -(1 << 30);  // rows above the owned ones that exist: ghost rows
    c.ghost.ghost_hi = s.whole.in_row0 + s.whole.in_rows > end ? end : (1 << 30);
    c.ghost.slots = kGateSlots;
    c.ghost.armed = true;
    const int rc = fn(s.whole, s.row0, s.rows_local);
    const bool taken = !c.ghost.armed;
    c.ghost = GhostGate();
    if (taken) return rc == TOPO_AMD_OK ? finish_gated(lean && device_gate, false) : rc;
    if (rc != TOPO_AMD_EUNSUP) return rc;  // (an error before anything was launched)
    return run_three(block, s, above, below, fn, true);
}


}  // namespace
}  // namespace topo

namespace {
// The class of the raster a shard belongs to, before the first launcher asks: a shard that is the whole raster is scanned
// like any whole block; a partial one uses what was declared for its owned rows, and declares it itself - collectively,
// every rank being in the same call on a buffer nobody has declared anything for since it was last written - when
// nothing was (a C caller that never heard of topo_amd_shard_classify gets the single GPU's bits too).
int ensure_shard_class(const Shard& s) {
    if (s.rows_local == s.owned.gny || declared_class(s.owned, nullptr)) return TOPO_AMD_OK;
    return topo_amd_shard_classify(s.owned.in, s.rows_local, s.row0, s.owned.gny, s.owned.nx);
}
}  // namespace

extern "C" {

int topo_amd_shard_tpi_std(float* block, int rows_local, int row0, int gny, int nx, int size,
                           float* tpi_out, float* std_out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    DiscRuns disc;
    TOPO_TRY(build_disc(size, &disc));
    const int above = -disc.dj_min, below = disc.dj_max;
    TOPO_TRY(shard_view(&block, above, below, "shard_tpi_std"));
    block += shard_view_offset(above, nx);
    Shard s = make_shard(block, rows_local, row0, gny, nx, above, below);
    TOPO_TRY(ensure_shard_class(s));
    ClassScope cls(s.owned);  // (the shard is the raster, or what ensure_shard_class found declared for its owned rows)
    forget_plane(tpi_out, rows_local, nx);
    forget_plane(std_out, rows_local, nx);
    // (tile rows of the first kernel: 48 / 60 for the ring kernels of STD, 60 for the marching TPI kernels, 64 for the TPI rings)
    const int tile_rows = std_out ? std_tile_rows(size) : (size <= 17 ? 64 : 60);
    return run_fused(block, s, above, below, tile_rows, [&](const Block& view, int o0, int on) {
        Block b = view;
        b.out_row0 = o0;
        b.out_rows = on;
        TOPO_TRY(check_block(b, above, below, "shard_tpi_std"));
        return launch_tpi_std(b, disc, shift(tpi_out, o0 - row0, nx), shift(std_out, o0 - row0, nx));
    });
}

int topo_amd_shard_gradient(float* block, int rows_local, int row0, int gny, int nx, double sigma,
                            double sig_ratio, int res_mode, const void* res_x, const void* res_y,
                            float* dx_out, float* dy_out, float* slope_out, float* aspect_out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    const int h = gradient_shard_halo(sigma, sig_ratio);  // == topo_amd_halo_rows(GRADIENT, sigma, sig_ratio)
    TOPO_TRY(shard_view(&block, h, h, "shard_gradient"));
    block += shard_view_offset(h, nx);
    Shard s = make_shard(block, rows_local, row0, gny, nx, h, h);
    TOPO_TRY(ensure_shard_class(s));
    ClassScope cls(s.owned);
    for (float* o : {dx_out, dy_out, slope_out, aspect_out}) forget_plane(o, rows_local, nx);
    return run_gated(block, s, h, h, false, [&](const Block& view, int o0, int on) {
        Block b = view;
        b.out_row0 = o0;
        b.out_rows = on;
        TOPO_TRY(check_block(b, h, h, "shard_gradient"));
        const void *rx = res_x, *ry = res_y;
        if (res_mode == TOPO_AMD_RES_2D) {
            rx = (const float*)res_x + (size_t)(o0 - row0) * nx;
            ry = (const float*)res_y + (size_t)(o0 - row0) * nx;
        }
        return launch_gradient(b, sigma, sig_ratio, res_mode, rx, ry, shift(dx_out, o0 - row0, nx),
                               shift(dy_out, o0 - row0, nx), shift(slope_out, o0 - row0, nx),
                               shift(aspect_out, o0 - row0, nx));
    });
}

int topo_amd_shard_sx(float* block, int rows_local, int row0, int gny, int nx, const int32_t* dj,
                      const int32_t* di, const double* dist, int n_off, int window, double height,
                      float* out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    int up = 0, down = 0;
    for (int n = 0; n < n_off; ++n) {
        if (std::isnan(dist[n])) continue;
        up = std::max(up, -dj[n]);
        down = std::max(down, dj[n]);
    }
    TOPO_TRY(shard_view(&block, up, down, "shard_sx"));
    block += shard_view_offset(up, nx);
    Shard s = make_shard(block, rows_local, row0, gny, nx, up, down);
    return run_gated(block, s, up, down, true, [&](const Block& view, int o0, int on) {
        Block b = view;
        b.out_row0 = o0;
        b.out_rows = on;
        TOPO_TRY(check_block(b, up, down, "shard_sx"));
        return launch_sx(b, dj, di, dist, n_off, window, height, shift(out, o0 - row0, nx));
    });
}

int topo_amd_shard_sx_multi(float* block, int rows_local, int row0, int gny, int nx, int n_az,
                            const int32_t* first, const int32_t* dj, const int32_t* di, const double* dist,
                            const int32_t* window, double height, float* const* outs) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(n_az >= 1 && first && dj && di && dist && window && outs, "shard_sx_multi: NULL argument");
    int up = 0, down = 0;
    sx_multi_reach(n_az, first, dj, dist, &up, &down);
    TOPO_TRY(shard_view(&block, up, down, "shard_sx_multi"));
    block += shard_view_offset(up, nx);
    Shard s = make_shard(block, rows_local, row0, gny, nx, up, down);
    std::vector<float*> moved(n_az);
    return run_overlapped(block, s, up, down, [&](const Block& view, int o0, int on) {
        Block b = view;
        b.out_row0 = o0;
        b.out_rows = on;
        TOPO_TRY(check_block(b, up, down, "shard_sx_multi"));
        for (int k = 0; k < n_az; ++k) moved[k] = shift(outs[k], o0 - row0, nx);
        return launch_sx_multi(b, n_az, first, dj, di, dist, window, height, moved.data());
    });
}

int topo_amd_shard_valley_ridge(float* block, int rows_local, int row0, int gny, int nx, const float* taps,
                                const int32_t* ksize, const float* angles, int n_angles, int n_planes,
                                float* norm_out, float* dir_out) {
    TOPO_ENTER();
    TOPO_TRY(require_ready());
    TOPO_REQUIRE(block && taps && ksize && angles && norm_out && dir_out && n_angles >= 1,
                 "shard_valley_ridge: NULL argument");
    int above = 0, below = 0;
    (void)valley_ridge_reach(ksize, n_angles, &above, &below);
    TOPO_TRY(shard_view(&block, above, below, "shard_valley_ridge"));
    block += shard_view_offset(above, nx);
    // the standardisation needs the mean and standard deviation of the WHOLE DEM (topo.py:427):
    // float64 moments about 0 of the owned rows, then the one all-reduce of the whole path.
    // On a DEM of whole metres the three numbers are exact integers (< 2^53), so every sharding
    // gets the same mean and std.
    double mom[3] = {(double)rows_local * (double)nx, 0.0, 0.0};
    TOPO_TRY(launch_moments(block + (size_t)above * nx, (size_t)rows_local * nx, 0.0, false, &mom[1], &mom[2]));
    if (g_comm.size > 1) {
        void* d_mom = nullptr;
        TOPO_TRY(workspace(0, 3 * sizeof(double), &d_mom));
        TOPO_HIP(hipMemcpyAsync(d_mom, mom, sizeof(mom), hipMemcpyHostToDevice, ctx().compute));
        TOPO_NCCL(ncclAllReduce(d_mom, d_mom, 3, ncclDouble, ncclSum, g_comm.comm, ctx().compute));
        TOPO_HIP(hipMemcpyAsync(mom, d_mom, sizeof(mom), hipMemcpyDeviceToHost, ctx().compute));
        TOPO_HIP(hipStreamSynchronize(ctx().compute));
    }
    const double mean = mom[1] / mom[0];
    double var = mom[2] / mom[0] - mean * mean;
    if (var < 0.0) var = 0.0;
    const double stdev = std::sqrt(var);
    Shard s = make_shard(block, rows_local, row0, gny, nx, above, below);
    return run_overlapped(block, s, above, below, [&](const Block& view, int o0, int on) {
        Block b = view;
        b.out_row0 = o0;
        b.out_rows = on;
        TOPO_TRY(check_block(b, above, below, "shard_valley_ridge"));
        return launch_valley_ridge(b, taps, ksize, angles, n_angles, n_planes, mean, stdev,
                                   shift(norm_out, o0 - row0, nx), shift(dir_out, o0 - row0, nx));
    });
}

}  // extern "C"
