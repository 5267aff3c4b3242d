// Shared declarations of libtopo_amd: context, error plumbing, launch helpers.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/topo_amd.h"

namespace topo {

// ---- error plumbing ---------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define TOPO_HIP(call)                                                                     \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            ::topo::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),       \
                              __FILE__, __LINE__);                                         \
            return TOPO_AMD_EHIP;                                                          \
        }                                                                                  \
    } while (0)

#define TOPO_REQUIRE(cond, ...)                                                            \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            ::topo::set_error(__VA_ARGS__);                                                \
            return TOPO_AMD_EINVAL;                                                        \
        }                                                                                  \
    } while (0)

#define TOPO_TRY(call)                                                                     \
    do {                                                                                   \
        int r_ = (call);                                                                   \
        if (r_ != TOPO_AMD_OK) return r_;                                                  \
    } while (0)

// Geometry of a row block inside the global DEM (see include/topo_amd.h).
struct Block {
    const float* in;
    int in_rows, in_row0, gny, nx;
    int out_row0, out_rows;
};

// ---- ghost-row gate (row shards) ----------------------------------------------------------------
// A sharded call runs ONE launch over the rows a shard owns: the interior rows first, then - behind this gate -
// the seam rows whose stencils read ghost rows.  The communication stream writes `epoch` into `*word` once the
// ghost rows of the exchange have landed (topo_amd_halo_exchange_start); the blocks that reach a seam part
// before that wait for it.  word == nullptr: no gate (the ghost rows are already final).
constexpr size_t kGateBytes = 16384;             // the gate word, then (from byte 256) one "gave up" byte per block of a launch
constexpr size_t kGateSlots = kGateBytes - 256;  // blocks a gated launch may have
struct Gate {
    const uint32_t* word;
    uint32_t epoch;
    uint32_t* timeouts;  // pinned host word: blocks that gave up waiting (a statistic: topo_amd_gate_giveups)
    uint32_t limit_ticks;  // s_memtime ticks (100 MHz) a block waits at most
    uint8_t* skipped;    // one byte per block of the launch: set by a block that gave up, cleared by the clean-up launch
    uint32_t* errors;    // pinned host word, lean mode (no clean-up launch): blocks whose wait ran out - an error
};

// The seam parts of a sharded call, taken up by the first launcher that supports parts (common.hpp: launchers
// that read `Context::seams` build one kernel argument per part; output pointers of a part are the main block's
// shifted by the difference of the first output rows).
struct Seams {
    int n = 0;          // seam parts behind the main block (0: an ordinary call)
    Block b[2];
    Gate gate{nullptr, 0, nullptr, 0, nullptr, nullptr};
    bool gate_armed = false;  // the first launch of the call waits at the gate; later ones are ordered behind it
};

// The other form of a sharded call, for kernels whose tiles are independent (Sx, the Gaussian passes): ONE launch over
// all the rows the shard owns, on the block with its ghost rows.  The launcher puts the tile rows that read ghost
// rows (global rows < ghost_lo or >= ghost_hi) at the end of the dispatch order, and their blocks wait at the gate.
// A launcher that cannot do that answers TOPO_AMD_EUNSUP BEFORE it launches anything (capi.hip then runs the
// interior and the seam strips as separate launches); one that can clears `armed`.
struct GhostGate {
    bool armed = false;
    Gate gate{nullptr, 0, nullptr, 0, nullptr, nullptr};
    int ghost_lo = 0, ghost_hi = 0;
    size_t slots = 0;  // bytes behind gate.skipped
};

// ---- per-process context (one process drives one GPU) -------------------------------------
struct Context {
    bool ready = false;
    int device = -1;
    int num_cu = 256;
    int reserve_cus = 0;  // CUs persistent kernels leave free while a ghost-row exchange is in flight
    Seams seams;          // set by run_fused (capi.hip) around the launchers of one sharded call
    GhostGate ghost;      // set by run_gated (capi.hip) around the launcher of one sharded call
    uint32_t* gate_word = nullptr;      // device word the communication stream writes the exchange epoch into
    uint32_t* gate_timeouts = nullptr;  // pinned host word (see Gate)
    uint32_t gate_epoch = 0;            // epoch of the last exchange started
    // How the seam parts wait (capi.hip, run_fused).  The library starts CAREFUL: a block waits a short while at a
    // closed gate, then leaves its seam tiles to a clean-up launch behind the exchange's event - correct whatever
    // RCCL's kernel needs to make progress.  After a few careful calls in which no block gave up (the exchange does
    // run next to the interior launch on this machine) it goes LEAN: no clean-up launch and no event wait on the
    // compute stream (20 us of a 600 us step), blocks wait up to TOPO_AMD_GATE_TIMEOUT_MS and a wait that runs out
    // is an error reported by the next synchronising call - after which the library is careful again for good.
    int gate_mode = 0;          // 0 careful (probing), 1 lean, 2 careful for good
    int gate_clean_calls = 0;   // consecutive careful calls that have completed without a block giving up
    hipEvent_t gate_probe = nullptr;  // recorded behind the last careful call
    bool gate_probe_pending = false;
    size_t lds_per_block = 65536;
    hipStream_t compute = nullptr;   // every kernel goes here
    hipStream_t comm = nullptr;      // RCCL ghost-row traffic
    // device planes of the host-buffer entry points: grow-only, handed out in the order a call asks for them and kept for
    // the next call (a hipMalloc / hipFree pair per GiB plane and call cost milliseconds, and copies out of freshly
    // mapped memory ran at half the link's rate: profiles/r05_host_pipeline.txt).  topo_amd_release_host_planes frees them.
    std::vector<void*> host_planes;
    std::vector<size_t> host_plane_bytes;
    hipStream_t up = nullptr, down = nullptr;  // the copies of a pipelined host-buffer call (capi.hip, run_pipelined; created on first use)
    std::vector<hipEvent_t> pipe_events;        // its events: uploaded chunk k, computed chunk k
    hipStream_t aux = nullptr;       // bandwidth-bound epilogues running next to matrix-core kernels (created on first use)
    hipEvent_t aux_ready[64] = {};    // compute -> aux, one per chunk
    hipEvent_t aux_done = nullptr;   // aux -> compute
    hipEvent_t halo_done = nullptr;  // comm -> compute dependency
    hipEvent_t input_ready = nullptr;  // compute -> comm dependency
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // grow-only device workspaces (never freed between calls: no hipMalloc in the hot path)
    void* ws[12] = {};
    size_t ws_bytes[12] = {};
    // small parameter tables (disc runs, gaussian taps, sx offsets, resolutions)
    void* tab[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t tab_bytes[6] = {0, 0, 0, 0, 0, 0};
};

Context& ctx();
int require_ready();

// ---- threading: one call at a time per context (SURVEY 8b "re-entrant per device context") -------------------------------
// The context owns things a call uses from its first to its last line - the grow-only workspaces `ws`, the device planes
// of the host-buffer entry points, the events and streams of their pipeline, the seam / gate state of a sharded call, the
// parameter-table staging - so every extern "C" entry point that touches the context holds the context's mutex for its
// whole duration (recursive: the host-buffer entry points call the device ones).  Calls of several threads therefore run
// one after the other, in the order they take the mutex; a device (`*_dev`, `topo_amd_shard_*`) call only ENQUEUES, so the
// mutex is held for microseconds there and the GPU work of the threads still queues up back to back on the compute
// stream.  The guard also binds the calling thread to the context's device (HIP's current device is per thread).
std::recursive_mutex& call_mutex();
struct CallGuard {
    std::lock_guard<std::recursive_mutex> lock;
    CallGuard();
    CallGuard(const CallGuard&) = delete;
    CallGuard& operator=(const CallGuard&) = delete;
};
#define TOPO_ENTER() ::topo::CallGuard call_guard_
int workspace(int slot, size_t bytes, void** out);       // device scratch, grow-only
int upload_table(int slot, const void* host, size_t bytes, void** out);  // async on compute

int check_block(const Block& b, int need_above, int need_below, const char* who);
// Several kernels launch one block row per DEM row (gridDim.y): 65 535 rows per launch at most.  The launch_*
// entry points therefore cut taller requests into row blocks of kMaxLaunchRows output rows (a multiple of every
// tile height in use: 32, 60, 64, 192) - row blocks give the single block's bits, so nobody sees the cut.
constexpr int kMaxLaunchRows = 65280;
inline int check_grid_rows(long rows, const char* who) {
    TOPO_REQUIRE(rows <= 65535, "%s: %ld rows in one call, the launch covers at most 65535 (gridDim.y): split the "
                 "request with out_row0 / out_rows", who, rows);
    return TOPO_AMD_OK;
}

// ---- disc geometry (host) ---------------------------------------------------------------
struct DiscRuns {
    int size = 0;
    int taps = 0;             // number of ones in the mask (centre included)
    int dj_min = 0, dj_max = 0;
    int di_min = 0, di_max = 0;
    int centre_dj = 0, centre_di = 0;      // offset of the tap TPI zeroes (topo.py:170)
    std::vector<int16_t> lo, hi;           // per dj in [dj_min, dj_max]: di run [lo, hi]
};
int build_disc(int size, DiscRuns* out);

// ---- kernel launchers (defined in the .hip files) -------------------------------------------
int launch_tpi_std(const Block& b, const DiscRuns& disc, float* tpi_out, float* std_out);
// size-specialised wave-shift kernels; TOPO_AMD_EUNSUP = not covered, use the generic kernel
int launch_disc_wave(const Block& b, int size, float* tpi_out, float* std_out);
bool disc_wave_covers(int size);  // a specialisation exists for this disc size
// Tile rows of the STD / TPI + STD ring kernels: the small discs take the build with staging waves apart from chain waves
// (disc_ring_impl.hpp, std_ring_spec_kernel: batches of 16 rows, map tiles of 48) up to 41 px, the others std_ring_kernel (12 and 60).
constexpr int kStdSpecMax = 41;
constexpr int std_tile_rows(int size) { return size >= 5 && size % 2 == 1 && size <= kStdSpecMax ? 48 : 60; }
// TPI of two small disc sizes from one pass over the DEM (disc_pair.hip); TOPO_AMD_EUNSUP = pair not covered
int launch_tpi_pair(const Block& b, int size_a, float* out_a, int size_b, float* out_b);
bool tpi_pair_covers(int size_a, int size_b);
// any size: float64 column prefix sums in HBM (slow, exact)
int launch_disc_big(const Block& b, const DiscRuns& disc, float* tpi_out, float* std_out);
int launch_gaussian(const Block& b, double sigma_y, double sigma_x, float* out, bool small_ok = true);
// smallest radius that takes the matrix-core Gaussian kernels (>= 8); below 16 they want 16 ghost rows
int mfma_min_radius(bool for_gradient);
int launch_sobel(const Block& b, float* dx_out, float* dy_out);
int launch_gradient(const Block& b, double sigma, double sig_ratio, int res_mode,
                    const void* res_x, const void* res_y, float* dx, float* dy, float* slope,
                    float* aspect);
int launch_sx(const Block& b, const int32_t* dj, const int32_t* di, const double* dist,
              int n_off, int window, double height, float* out);
// several azimuth sectors in one pass: sector a owns entries [first[a], first[a+1]) of the tables
int launch_sx_multi(const Block& b, int n_az, const int32_t* first, const int32_t* dj, const int32_t* di,
                    const double* dist, const int32_t* window, double height, float* const* outs);
int launch_synth(float* out, int rows, int row0, int nx, uint32_t seed, bool integer_valued);
// valley / ridge index (valley.hip): taps = per angle ksize^2 x 4 floats (plane sums, flipped)
// the same by FFT, for kernels of any size (valley_fft.hip)
int launch_valley_ridge_fft(const Block& b, const float* taps, const int32_t* ksize, const float* angles,
                            int n_angles, int n_planes, int kmax, double mean, double stdev, float* norm_out,
                            float* dir_out);
void valley_fft_release();  // destroys the cached FFT plans (topo_amd_shutdown)
int valley_ridge_reach(const int32_t* ksize, int n_angles, int* above, int* below);
void note_valley_route(int route);  // what topo_amd_valley_route reports for the calling thread (capi.hip)
// the same on the matrix pipe for rotated kernels of up to kValleyMfmaMaxKernel cells a side with at most 240 cells that hold a
// tap at any angle (valley_mfma.hip; *done = 0: not such a case, nothing launched); leaves the pixels it cannot do marked
// norm = -1 and their tiles (kValleyMfmaTileRows x 64, anchored at out_row0) flagged
constexpr int kValleyMfmaMaxKernel = 25;     // ... with a pixel tile's operands in registers
constexpr int kValleyStreamMaxKernel = 120;  // ... folded with the operands streamed (point-symmetric tables)
constexpr int kValleyMfmaTileRows = 32;
int launch_valley_ridge_mfma(const Block& b, const float* taps, const int32_t* ksize, const float* angles, int n_angles,
                             int n_planes, int kmax, double mean, double stdev, float* norm_out, float* dir_out,
                             const int** flags_out, int* flag_cols, int* done);
int launch_mean_std(const float* in, size_t count, double* mean, double* stdev);
int launch_moments(const float* in, size_t count, double pivot, bool pivot_is_first_sample, double* sum, double* sumsq);
int launch_valley_ridge(const Block& b, const float* taps, const int32_t* ksize, const float* angles, int n_angles,
                        int n_planes, double mean, double stdev, float* norm_out, float* dir_out);

int gaussian_radius(double sigma);

// ---- the raster class: what kernel routing may know about the WHOLE raster -------------------------------------------
// Two kernel choices change the last bits of a result (never its correctness): the Gaussian / gradient of a raster whose
// ordinary values lie beyond the f16 matrix-core kernels' range runs on the vector-ALU kernels, and TPI alone on
// fractional elevations sums x in units of 2^-k m with k taken from the raster's value range.  Both are decided from a
// sample of the WHOLE raster on a lattice of the GLOBAL grid (topo_amd_raster_scan_dev), never from the block of one call:
// a block that is the whole raster is scanned by the library (remembered per buffer until the library writes the buffer
// or topo_amd_dem_changed names it); the host-buffer entry points scan the caller's array; a partial row block uses what
// was declared FOR ITS MEMORY (topo_amd_raster_class_set / _from_scan: keyed by the block's device rows and the raster's
// shape, dropped when the library writes or frees them - never a property of a thread or of the process); the
// topo_amd_shard_* calls declare it themselves on first use (topo_amd_shard_classify, collective); a partial block nobody
// declared anything for is an ordinary DEM in whole metres.  So every row block of a raster takes the whole raster's kernels.
struct RasterClass {
    bool large = false;   // more than a quarter of the lattice samples are finite and beyond +-1e5
    float lo = 0.0f, hi = 4096.0f;  // smallest / largest ordinary lattice sample (finite, within +-2^18); lo > hi: none seen
    float frac_share = 0.0f;        // share of the lattice samples with a fractional part (TIME only: which disc kernels go first)
};
RasterClass current_class();  // of the call in flight on this thread (capi.hip sets it around the launchers)

// What the library remembers about the blocks it has seen (keyed by block pointer, rows and width; a few entries) - TIME
// only, never bits: the share of tiles with fractional elevations in one block's run of the last TPI call, written by
// that block into pinned host memory (no stream operation, no synchronisation; a call reads what the last finished call
// left).  A DEM remembered as mostly fractional starts with the kernel that suits it (tpi_scaled_march_kernel<TAKE_ALL>:
// the same bits as the two-launch route, pixel by pixel).  An entry is dropped whenever the library writes or frees memory
// that overlaps its block (dem_memo_forget: uploads, copies, memset, the synthetic DEM, every output plane, the
// workspaces, topo_amd_free) and by topo_amd_dem_changed.
uint32_t* dem_memo_report(const Block& b);           // the pinned words {tiles, fractional tiles} of this DEM's entry
bool dem_memo_mostly_fractional(const Block& b);
// The same memory for the Gaussian's fused matrix-core kernel: the word it sets when it stages a sample that is not a
// plain finite one (non-finite, or beyond 1e5 in magnitude), and what that word said after the last finished call
// (the two-pass kernels give the fused kernel's bits).
uint32_t* dem_memo_wild_word(const Block& b);
bool dem_memo_wild(const Block& b);
void dem_memo_forget(const void* p, size_t bytes);

}  // namespace topo
