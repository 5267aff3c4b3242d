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
// use the generic LDS kernel in disc.hip.  Persistent blocks walk the tile list (vertical
// neighbours first, cut into XCD-contiguous runs so that the ghost rows two tiles share meet in
// one L2).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "disc_runs.hpp"
#include "gate.hpp"

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
    uint32_t* scratch;  // per-block planes for the per-row sums between the passes
    uint8_t* defer;     // per tile: 1 = left by the marching kernels for the general kernel; null = no split
    int32_t* sums;      // marching kernels: exact sum of trunc(x) over the disc per pixel, laid out like the outputs
    int map_th;         // general kernel: tile height of the geometry p.defer was built for (0 = its own)
    int map_tiles_y;    //                 and its number of tile rows
    int map_tw;         //                 and its strip width when that differs too (0 = its own)
    float* tpi2;        // ring kernel for a pair of disc sizes: TPI of the smaller disc
    int scaled;         // TPI alone, tiles with fractional elevations: 1 = the scaled one-chain route (tpi_scaled_march_kernel)
    uint32_t* report;   // pinned host words {tiles, tiles with fractional samples} of ONE block's run (dem_memo), or nullptr
    float unit;         // the scaled route: samples are summed as rint(unit x), unit = 2^8 ... 2^16 (scaled_unit)
    uint64_t* rowmask;  // per tile of state kTileGeneralRows: the rows (bit = row of the tile) the general kernel is to write
    uint32_t* sums2;    // fractional STD by three marching passes: sum of (trunc(x) - c)^2 per pixel, laid out like the outputs ...
    int32_t* tile_c;    // ... and the offset c of every tile that has them
};

// One launch, several row blocks ("parts").  An ordinary call has one part.  A sharded call (capi.hip, run_fused)
// has the rows whose stencils stay inside the shard as part 0 and the seam strips - the rows that read ghost rows -
// as parts 1 and 2, each with the block view, the output rows and the slices of the workspaces that the separate
// launches of rounds 1-3 gave it (so the bits cannot differ).  Every persistent block walks its share of part 0,
// waits at the gate (gate.hpp) and walks its share of the seam parts: the seams ride in the tail of the interior
// launch instead of two under-filled launches behind an event.  `shift` rotates the block numbering of a part,
// so that the few tiles of a seam part go to the blocks at the end of the grid, whose interior runs are the
// shortest ones.
constexpr int kMaxParts = 3;
// How a part's tile list is dealt to the blocks: block vb (after the XCD-aware renumbering) acts as block
// (vb + shift) % nb and takes a contiguous run of the strip-major tile list; the runs differ by at most one tile
// (deal_parts), none is empty while there are as many tiles as blocks.
struct PartRun {
    int shift;  // block vb acts as block (vb + shift) % nb
    int base;   // tiles of a run ...
    int b1, b2, mid;  // ... plus `mid` (+1 or -1) for the blocks b1 <= vb < b2
    __host__ __device__ int first(int vb) const { return vb * base + mid * (vb <= b1 ? 0 : (vb < b2 ? vb - b1 : b2 - b1)); }
    __host__ __device__ int count(int vb) const { return base + (vb >= b1 && vb < b2 ? mid : 0); }
};
struct WaveParts {
    WaveArgs a[kMaxParts];
    int tiles_y[kMaxParts];
    PartRun run[kMaxParts];
    int n;
    int cleanup;               // the clean-up launch behind the exchange's event: seam parts of the blocks that gave up
    Gate gate;
};

// The kernels' common frame.  (1) Which block am I: workgroups go to the XCDs round-robin, so numbering the blocks
// of a grid of whole rounds XCD by XCD (vb) puts neighbouring runs of the tile list behind one L2.  (2) The parts:
// part 0, then - behind the gate - the seam parts; a block that gives up at the gate marks itself, and the clean-up
// launch (same grid, same dealing) does the seam parts of the marked blocks.
// body: `void body(const WaveArgs&, int tiles_x, int tiles_y, PartRun deal, int vb0, int nb)`.  (One copy of the
// body in a loop, the parts indexed in the kernel arguments: scalar loads, no scratch.)
// An ordinary call is one part: the kernel with the arguments of rounds 1-3 (the frame's loop state costs the ring
// kernels at the register limit dear: std_ring_kernel<65, true> spilt 109 VGPRs through it).
#define TOPO_RUN_ONE(body)                                                                              \
    const int nb_ = (int)gridDim.x;                                                                     \
    const int vb_ = (nb_ & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nb_ >> 3) + (int)(blockIdx.x >> 3); \
    body(p, tiles_x, tiles_y, deal, vb_, nb_);
#define TOPO_RUN_PARTS(body)                                                                            \
    const int nb_ = (int)gridDim.x;                                                                     \
    const int vb_ = (nb_ & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nb_ >> 3) + (int)(blockIdx.x >> 3); \
    if (ps.cleanup) {                                                                                   \
        if (ps.gate.skipped[blockIdx.x] == 0) return;                                                   \
        __syncthreads();                                                                                \
        if (threadIdx.x == 0) ps.gate.skipped[blockIdx.x] = 0;                                          \
    } else {                                                                                            \
        /* part 0 - nearly all of the work - is a copy of the body with its arguments at fixed offsets, like the */ \
        /* ordinary kernel's; the seam parts share a second copy in a loop (what that loop keeps alive cost the */ \
        /* ring kernels at the register limit 8 % when part 0 went through it too) */               \
        body(ps.a[0], tiles_x, ps.tiles_y[0], ps.run[0], vb_, nb_);                                     \
        if (!gate_wait(ps.gate, blockIdx.x) && ps.gate.errors == nullptr) return;                       \
    }                                                                                                   \
    _Pragma("unroll 1") for (int part_ = 1; part_ < ps.n; ++part_) {                                    \
        __syncthreads();                                                                                \
        body(ps.a[part_], tiles_x, ps.tiles_y[part_], ps.run[part_], vb_, nb_);                         \
    }

// One copy of the body for all parts, in a loop: the kernels of the slower paths (the general kernel, the fraction and
// scaled passes, the second marching kernel of STD) are launched through this form only - an ordinary call is ps.n == 1
// with no gate - which is a third of the code of "one kernel for ordinary calls + a parts kernel with two copies".
#define TOPO_RUN_PARTS_LOOP(body)                                                                       \
    const int nb_ = (int)gridDim.x;                                                                     \
    const int vb_ = (nb_ & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nb_ >> 3) + (int)(blockIdx.x >> 3); \
    int first_part_ = 0;                                                                                \
    if (ps.cleanup) {                                                                                   \
        if (ps.gate.skipped[blockIdx.x] == 0) return;                                                   \
        __syncthreads();                                                                                \
        if (threadIdx.x == 0) ps.gate.skipped[blockIdx.x] = 0;                                          \
        first_part_ = 1;                                                                                \
    }                                                                                                   \
    _Pragma("unroll 1") for (int part_ = first_part_; part_ < ps.n; ++part_) {                          \
        if (part_ == 1 && !ps.cleanup) {                                                                \
            if (!gate_wait(ps.gate, blockIdx.x) && ps.gate.errors == nullptr) return;                   \
        }                                                                                               \
        if (part_ > 0) __syncthreads();                                                                 \
        body(ps.a[part_], tiles_x, ps.tiles_y[part_], ps.run[part_], vb_, nb_);                         \
    }

// Fills `ps` for the block `b` (arguments `a0`: the launcher's, with the output pointers of `b`) and for the seam
// parts of the sharded call in flight, if any.  tile_h x strip_w: the kernel's tile geometry; want_defer /
// want_sums: slices of workspaces 8 (one byte per tile) and 9 (an int32 per output pixel).  map_th / map_tw: the
// general kernel reading the tile map of a marching launch with another geometry (rows of map_th, strips of
// map_tw; 0 = its own).  Returns the number of strips and the tiles of part 0 (what the grid is sized by).
inline int make_parts(const Block& b, const WaveArgs& a0, int tile_h, int strip_w, bool want_defer, bool want_sums,
                      WaveParts* ps, int* tiles_x_out, long* ntiles0, int map_th = 0, int map_tw = 0) {
    Context& c = ctx();
    const int n = 1 + c.seams.n;
    const int tiles_x = (b.nx + strip_w - 1) / strip_w;
    const bool other_map = map_th != 0 && (map_th != tile_h || map_tw != 0);
    const int mtw = (other_map && map_tw != strip_w) ? map_tw : 0;
    const int map_tiles_x = mtw ? (b.nx + mtw - 1) / mtw : tiles_x;
    size_t defer_off[kMaxParts + 1] = {0}, sums_off[kMaxParts + 1] = {0};
    for (int k = 0; k < n; ++k) {
        const Block& bk = k == 0 ? b : c.seams.b[k - 1];
        WaveArgs& a = ps->a[k];
        a = a0;
        a.in = bk.in;
        a.in_rows = bk.in_rows;
        a.in_row0 = bk.in_row0;
        a.out_row0 = bk.out_row0;
        a.out_rows = bk.out_rows;
        if (k > 0) a.report = nullptr;  // (the DEM memo hears from a block of part 0, not from a run of one seam tile)
        const ptrdiff_t moved = (ptrdiff_t)(bk.out_row0 - b.out_row0) * b.nx;
        if (a.tpi) a.tpi += moved;
        if (a.sd) a.sd += moved;
        if (a.tpi2) a.tpi2 += moved;
        ps->tiles_y[k] = (bk.out_row0 + bk.out_rows - 1) / tile_h - bk.out_row0 / tile_h + 1;
        long map_tiles = (long)tiles_x * ps->tiles_y[k];
        if (other_map) {
            a.map_th = map_th;
            a.map_tiles_y = (bk.out_row0 + bk.out_rows - 1) / map_th - bk.out_row0 / map_th + 1;
            a.map_tw = mtw;
            map_tiles = (long)map_tiles_x * a.map_tiles_y;
        }
        defer_off[k + 1] = defer_off[k] + (size_t)map_tiles;
        sums_off[k + 1] = sums_off[k] + (size_t)bk.out_rows * b.nx;
        ps->run[k] = PartRun{0, 0, 0};
    }
    for (int k = n; k < kMaxParts; ++k) {
        ps->a[k] = ps->a[0];
        ps->tiles_y[k] = 0;
        ps->run[k] = PartRun{0, 0, 0};
    }
    if (want_defer) {
        void* defer = nullptr;  // the same sizes in every launch of a group, so the same slices of one allocation
        const size_t masks_at = (defer_off[n] + 15) & ~(size_t)15;  // one uint64 per tile behind the state bytes, then one int32
        const size_t cs_at = masks_at + defer_off[n] * sizeof(uint64_t);
        TOPO_TRY(workspace(8, cs_at + defer_off[n] * sizeof(int32_t), &defer));
        for (int k = 0; k < n; ++k) {
            ps->a[k].defer = (uint8_t*)defer + defer_off[k];
            ps->a[k].rowmask = (uint64_t*)((uint8_t*)defer + masks_at) + defer_off[k];
            ps->a[k].tile_c = (int32_t*)((uint8_t*)defer + cs_at) + defer_off[k];
        }
    }
    if (want_sums) {
        void* sums = nullptr;
        TOPO_TRY(workspace(9, 2 * sums_off[n] * sizeof(int32_t), &sums));  // (the second half: WaveArgs::sums2)
        for (int k = 0; k < n; ++k) {
            ps->a[k].sums = (int32_t*)sums + sums_off[k];
            ps->a[k].sums2 = (uint32_t*)sums + sums_off[n] + sums_off[k];
        }
    }
    ps->n = n;
    ps->gate = Gate{nullptr, 0, nullptr, 0, nullptr, nullptr};
    ps->cleanup = 0;
    if (n > 1 && c.seams.gate_armed) {
        ps->gate = c.seams.gate;
        c.seams.gate_armed = false;
    }
    *tiles_x_out = tiles_x;
    *ntiles0 = (long)tiles_x * ps->tiles_y[0];
    return TOPO_AMD_OK;
}

// Deals the tiles of every part to the `grid` blocks of the launch (PartRun).  An ordinary call: runs of
// ceil(ntiles / grid) tiles.  With seam parts: the seam tiles go round the ring of blocks one each, starting at block
// 0 - part 1 to blocks 0 ... n1 - 1, part 2 to the blocks behind them, wrapping - so the first (n1 + n2) % grid blocks
// hold one seam tile more than the others, and exactly those blocks get an interior run one tile shorter: every
// block ends up with the same number of tiles (a seam tile - a full staging for half a tile of rows - costs about
// what a marched interior tile costs).  The launches of one group use the same grid, hence the same runs.
inline void deal_parts(WaveParts* ps, int tiles_x, long grid, int blocks_per_cu) {
    (void)blocks_per_cu;
    const int nb = (int)grid;
    long seam = 0;
    bool ring = true;  // every seam part fits once round the blocks
    for (int k = 1; k < ps->n; ++k) {
        const long nk = (long)tiles_x * ps->tiles_y[k];
        ring = ring && nk <= nb;
        seam += nk;
    }
    const long n0 = (long)tiles_x * ps->tiles_y[0];
    // Every block should end up with floor or ceil of (n0 + seam) / nb tiles in all.  The seam tiles go round the ring of
    // blocks one each, starting at block 0 (below), so the first `light` = seam % nb blocks hold one seam tile more than
    // the others; with `rem` = (n0 + seam) % nb blocks owed the extra tile of the division, the interior run of block b
    // has base + [b < rem] - [b < light] tiles: base everywhere except between the two marks.
    const int light = (ps->n > 1 && ring) ? (int)(seam % nb) : 0;
    const long total = n0 + (light ? seam : 0);
    const int rem = (int)(total % nb);
    int base = (int)(total / nb - (light ? seam / nb : 0));
    if (base < 1 || (base == 1 && light > rem)) {  // fewer tiles than blocks (or nearly): the plain split of part 0
        ps->run[0] = PartRun{0, (int)(n0 / nb), 0, (int)(n0 % nb), 1};
    } else {
        ps->run[0] = PartRun{0, base, std::min(light, rem), std::max(light, rem), rem >= light ? 1 : -1};
    }
    long before = 0;
    for (int k = 1; k < ps->n; ++k) {
        const long nk = (long)tiles_x * ps->tiles_y[k];
        // block vb acts as block (vb - before) mod nb: tile t of this part lands on block (before + t) mod nb
        ps->run[k] = PartRun{(int)((nb - before % nb) % nb), (int)(nk / nb), 0, (int)(nk % nb), 1};
        before += nk;
    }
}

template <typename T>
struct alignas(16) Vec4 {
    T v[4];
};

#define DPP_WAVE_SHL1 0x130  // lane i takes lane i + 1; lane 63 takes 0 (bound_ctrl)

#ifdef HOP_BPERMUTE
// lab variant (tools/ubench/tpi_lab.hip -DHOP_BPERMUTE): the lane shift through the LDS crossbar (ds_bpermute_b32, no
// LDS memory) instead of a DPP operand - takes the hop off the vector ALU (where every DPP form issues at half rate)
// and puts it on the LDS pipe.  Same bits; TPI 67 px 4.42 -> 7.30 ms, STD 8.85 -> 11.03 ms on 32768^2: the crossbar's
// latency sits in the 18-hop dependent chain and its issue competes with the 44 ds_read_b128 of a row.
__device__ __forceinline__ int hop_addr() {
    return (int)(((__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) + 1u) & 63u) << 2);
}
__device__ __forceinline__ float hop(float x) { return __int_as_float(__builtin_amdgcn_ds_bpermute(hop_addr(), __float_as_int(x))); }
__device__ __forceinline__ int hop(int x) { return __builtin_amdgcn_ds_bpermute(hop_addr(), x); }
__device__ __forceinline__ uint32_t hop(uint32_t x) { return (uint32_t)__builtin_amdgcn_ds_bpermute(hop_addr(), (int)x); }
#else
__device__ __forceinline__ float hop(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), DPP_WAVE_SHL1, 0xf, 0xf, true));
}
__device__ __forceinline__ int hop(int x) {
    return __builtin_amdgcn_update_dpp(0, x, DPP_WAVE_SHL1, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t hop(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, DPP_WAVE_SHL1, 0xf, 0xf, true);
}
#endif

// Disc sums of output row jj (tile-relative) for the NC output columns that end up in this
// lane: lane l receives the sums of staged columns NC * (l - D_LO) + t, valid for l < NVL.
// T is the type of the prefix sums (float, int32, uint32 with wrap-around).  HALF selects what
// of each uint32 column sum enters the chain: 0 all of it, 1 its low 16 bits, 2 its high 16 bits
// (two 16-bit chains give an exact 48-bit total where one uint32 chain could overflow).
// PIPE pins the order "read the prefix rows of run r+1, then subtract those of run r": left to
// itself the scheduler does that in the kernels with registers to spare, and in the ones at the
// 168-VGPR limit it waits for every pair of reads before issuing the next (21 exposed LDS
// latencies per row).
// CHAIN_PRIO (default on; used by the ring kernels, disc_ring_impl.hpp): a wave lowers its own issue priority as it
// gets through its row.  The hardware serves the oldest wave of a SIMD first, so in a kernel whose waves each take
// ONE row between two barriers the three waves that share a SIMD finish one after the other and the last one runs
// alone, with nothing to hide its LDS latencies behind; with the laggard always ahead in priority they advance
// together (std_ring_kernel<67>: 10.19 -> 9.37 ms, TPI + STD 10.61 -> 9.85 ms on the 32768^2 bench DEM).  The
// marching kernels, whose waves draw rows from a queue (MARCH_DYN_ROWS), lose with it (4.44 -> 4.76 ms at 67 px):
// wave_disc_sum leaves the priority alone.
#ifndef CHAIN_PRIO
#define CHAIN_PRIO 1
#endif
#if CHAIN_PRIO
#define CHAIN_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define CHAIN_SETPRIO(x)
#endif

template <int SIZE, typename T, int HALF = 0, bool PIPE = false>
__device__ __forceinline__ void wave_disc_sum(const T* Q, int jj, int lane, T (&acc)[NC]) {
    using ACC = T;
    using G = Geo<SIZE>;
    constexpr int NR = G::T.num_runs;
    T cv[NR][NC];
    const T* col = Q + lane * NC;
    if (PIPE) {
        typedef T t4 __attribute__((ext_vector_type(4)));
        t4 top = *reinterpret_cast<const t4*>(col + (jj + G::T.run_hi[0] - G::T.off_min + 1) * ROWW);
        t4 bot = *reinterpret_cast<const t4*>(col + (jj + G::T.run_lo[0] - G::T.off_min) * ROWW);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            t4 ntop = top, nbot = bot;
            if (r + 1 < NR) {
                ntop = *reinterpret_cast<const t4*>(col + (jj + G::T.run_hi[r + 1] - G::T.off_min + 1) * ROWW);
                nbot = *reinterpret_cast<const t4*>(col + (jj + G::T.run_lo[r + 1] - G::T.off_min) * ROWW);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NC; ++s) {
                T d = top[s] - bot[s];
                if (HALF == 1) d = (T)((uint32_t)d & 0xffffu);
                if (HALF == 2) d = (T)((uint32_t)d >> 16);
                cv[r][s] = d;
            }
            __builtin_amdgcn_sched_barrier(0);
            top = ntop;
            bot = nbot;
        }
    } else {
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

// ---- TPI, STD, and TPI + STD fused: exact integer chains -----------------------------------------
// With u = trunc(x) - c, f = x - trunc(x) and the sums taken over the in-domain taps (m of them):
//   s1 = sum x          = Su + Sf + c m
//   s2 = sum trunc(x)^2 = Su2 + 2 c Su + c^2 m            (the int32 quirk of topo.py:300)
//   STD = sqrt(max(0, (s2 - s1^2/n) / (n-1))),   TPI = x - (s1 - x_ctr) / (n-1)
// Su and Su2 are integer sums (int32 / uint32 prefix sums cannot round; the uint32 chain switches
// to two 16-bit half chains when a tile's |u| would let 3409 u^2 pass 2^32), Sf is the sum of the
// fractional parts in integer units of 2^-16 m (exact as well) and exists only on fractional DEMs.  Tiles holding non-finite or absurd samples (|u| so
// large that one 67-row column sum of u^2 passes 2^32, e.g. -9999 nodata next to real terrain)
// run float chains on a = x - c and (trunc(x) - c)^2 instead, so NaN propagates and nothing wraps.
// kStLimb: one of the three kinds below, picked at run time by aux bits 4-5 (0 kStInd, 1 kStUL, 2 kStU2L) - the limb path runs its
// six passes through ONE copy of the staging and chain code (it is the slow path; six inlined copies per general kernel and disc
// size were a fifth of the library's code and of its build time)
enum Stage { kStU = 0, kStU2 = 1, kStF = 2, kStS = 5, kStInd = 6, kStUL = 7, kStU2L = 8, kStLimb = 9 };
// The general kernel on tiles the 32-bit integer chains cannot hold - samples beyond +-2^18 (a raster in millimetres), more
// relief than the chains of u^2 take (nodata next to terrain), non-finite samples.  Still exact integers, in LIMBS:
//   kStInd  per sample (not ordinary: non-finite or |trunc(x)| > 2^18) + 2^16 (missing: non-finite or |trunc(x)| >= 2^24): the
//           chain counts both kinds per disc;
//   kStUL   u = trunc(x) - c in two limbs (low 16 bits, the rest), missing samples as u = 0;
//   kStU2L  u^2 (< 2^50) in limbs of 16 bits, the last one taking what is left (< 2^18).
// Every limb sum over a disc fits 32 bits, the limbs add up to exact 64-bit sums, and the finalisation works on exact
// integers (128-bit where it must), so the result does not depend on c, on the tile or on what else the block has in view:
// a pixel whose disc holds a missing sample is NaN (exactly those pixels: the footprint of the disc), a pixel whose disc holds
// ordinary samples only gets the bits every other kernel gives it, and the rest - discs with samples beyond 2^18 - are the
// float64 rounding of the exact variance / mean.
constexpr float kMissingLim = 16777216.0f;  // 2^24 (where float32 stops holding every integer): |u| < 2^25 whatever c is, u^2 < 2^50, 8192 of them < 2^63
constexpr int kScratchPlanes = 6;            // planes of per-row sums a block of the general kernel keeps between its passes
// kStS: x in units of 1 / unit m (unit = 2^8 ... 2^16, one value per call: WaveArgs::unit), rounded to the nearest integer - an
// ABSOLUTE quantisation (no offset), so the value does not depend on which tile, run or row block stages the sample
// (tpi_scaled_march_kernel).  stage_value<kStS> takes the unit in the place of the offset c.

// sqrt(max(0, (s2 - s1^2/n) / (n-1))) from the float64 sums.  The variance stays in float64 (the
// difference cancels a few hundred-fold), but 1/n and 1/(n-1) are multiplied in and the root is
// taken in float32: an IEEE float64 division and square root per pixel cost more than the 67-step
// chain (159 f64 instructions per wave-row against 24 for TPI).  Every wave-shift kernel uses
// this one function, so they agree bit for bit.
__device__ __forceinline__ float std_from_sums(double s1, double s2, double inv_n, double inv_nm1) {
    double var = (s2 - s1 * s1 * inv_n) * inv_nm1;
    if (var < 0.0) var = 0.0;  // keeps NaN, like np.clip
    return sqrtf((float)var);
}
// The same for a pixel whose n taps are all inside the DEM, on a tile of the integer path:
// n s2 - s1^2 = n Su2 - Su^2 whatever offset c the sums were taken with, a non-negative integer below 2^62 that
// 64-bit integer arithmetic forms exactly (one v_mad_u64_u32 and one v_mad_i64_i32 when Su2 fits 32 bits), hence
// identical bits from kernels that used different offsets.  It goes to float32 as high word x 2^32 + low word in one
// fma (the two words are each rounded to float32 when they exceed 2^24, then the fma rounds: <= 1.8e-7 relative), is
// scaled by the float32 value of 1 / (n (n - 1)) (two more roundings: <= 1.2e-7) and rooted with the hardware's
// v_sqrt_f32 (the root halves the relative error of its argument and adds 1 ulp): 3e-7 relative in all on STD, against
// the 1e-4 of the contract (tests: test_std_vs_reference), in 8 instructions where the float64 form with a
// correctly rounded sqrtf took about 20 per pixel (a tenth of std_ring_kernel's vector instructions).  Every
// wave-shift and ring kernel uses this one function for such pixels, so they agree bit for bit.
__device__ __forceinline__ float std_from_int_sums(int su, uint64_t su2, uint32_t n, float inv_nn1) {
    const uint64_t num = (uint64_t)n * su2 - (uint64_t)((int64_t)su * (int64_t)su);
    const float f = fmaf((float)(uint32_t)(num >> 32), 4294967296.0f, (float)(uint32_t)num);
    return __builtin_amdgcn_sqrtf(f * inv_nn1);
}
// The same for a pixel of whole metres some of whose taps lie outside the DEM (they read 0, and n is still the full tap
// count): with m taps inside, T = sum trunc(x) = Su + c m and S2 = sum trunc(x)^2 = Su2 + 2 c Su + c^2 m are exact 64-bit
// integers (|T| < 2^31, S2 < 2^49 for ordinary samples), n S2 - T^2 is the exact numerator whatever c the sums were taken
// with, and the tail is std_from_int_sums'.  Round 5: every kernel finalises such pixels with THIS function (they had the
// float64 form: ~100 instructions per pixel, which made the tiles at the DEM's border several times dearer than the others).
__device__ __forceinline__ float std_from_border_sums(int64_t su, uint64_t su2, int ci, int m, uint32_t n, float inv_nn1) {
    const int64_t T = su + (int64_t)ci * m;
    const int64_t S2 = (int64_t)su2 + 2 * (int64_t)ci * su + (int64_t)ci * ci * m;
    const uint64_t num = (uint64_t)n * (uint64_t)S2 - (uint64_t)(T * T);
    const float f = fmaf((float)(uint32_t)(num >> 32), 4294967296.0f, (float)(uint32_t)num);
    return __builtin_amdgcn_sqrtf(f * inv_nn1);
}
// The limb path's finalisation of a pixel whose disc holds samples beyond 2^18: everything is an exact integer -
//   T = sum trunc(x) = Su + c m,   S2 = sum trunc(x)^2 = Su2 + 2 c Su + c^2 m   (m in-domain taps of n),
//   F = sum of the fractional parts = Sg / 2^16,   n s2 - s1^2 = (n S2 - T^2) - 2 T F - F^2
// - so 2^32 (n s2 - s1^2) is formed exactly in 128 bits (< 2^110), goes to float64 in one step (two conversions and an
// addition: a fixed function of an exact integer) and STD = sqrt(max(0, that / (2^32 n (n - 1)))) with the tail of
// std_from_sums.  No c, no tile, no block view enters the value.
__device__ __forceinline__ double int128_to_double(__int128 v) {
    const bool neg = v < 0;
    const unsigned __int128 mag = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
    const double d = (double)(uint64_t)(mag >> 64) * 18446744073709551616.0 + (double)(uint64_t)mag;
    return neg ? -d : d;
}
__device__ __forceinline__ float std_from_exact_sums(int64_t T, __int128 S2, int sg, int n, double inv_nn1) {
    const __int128 A = (__int128)n * S2 - (__int128)T * T;
    const __int128 num = (A << 32) - (((__int128)T * sg) << 17) - (__int128)sg * sg;
    double var = int128_to_double(num) * (1.0 / 4294967296.0) * inv_nn1;
    if (var < 0.0) var = 0.0;
    return sqrtf((float)var);
}
// kTileFloat (the name is round 1's): samples beyond 2^18 or more relief than the 32-bit chains hold - the limb passes.
// kTileMissing: a missing sample (non-finite, beyond +-2^24) - staged as u = 0 by every pass and counted by one more (kStInd),
// the pixels whose discs hold one are NaN; the tile's other passes are the ordinary ones unless kTileFloat is up as well.
enum TileFlags { kTileFrac = 1, kTileWide = 2, kTileFloat = 4, kTileMissing = 8 };
// The per-tile byte map (WaveArgs::defer).  kTileDone: finished; kTileGeneral: left to the general kernel (disc_wave_kernel);
// kNeedsFraction: its sums of trunc(x) are done, the fraction / scaled pass finishes it.
// kTileGeneralRows: the general kernel writes only the rows named in WaveArgs::rowmask (the scaled route left them: their
// own windows hold more relief than its unwrapping takes; the other rows of the tile are done)
// kTileGeneralWide: the general kernel's like kTileGeneral (odd), left by std_ring_kernel for its RANGE (more relief than the chain
// of squares takes) and not for fractional samples: nothing for that kernel's second pass (kStdBoth), which has the same limit
enum TileState : uint8_t { kTileDone = 0, kTileGeneral = 1, kNeedsFraction = 2, kTileGeneralRows = 3, kTileGeneralWide = 5 };

// aux: kStUL the limb (0 / 1); kStU2L the limb (bits 0-1) and "the last limb: all the bits that are left" (bit 2)
template <int WHAT>
__device__ __forceinline__ uint32_t stage_value(float x, float c, int ci, int aux = 0) {
    if (WHAT == kStS) return (uint32_t)(int)rintf(x * c);
    const float t = truncf(x);
    if (WHAT == kStInd || WHAT == kStUL || WHAT == kStU2L || WHAT == kStLimb) {
        const int kind = WHAT == kStLimb ? ((aux >> 4) & 3) : (WHAT == kStInd ? 0 : (WHAT == kStUL ? 1 : 2));
        const bool missing = !(fabsf(t) < kMissingLim);  // (NaN: missing)
        if (kind == 0) return (missing ? 65536u : 0u) | ((missing || fabsf(t) > 262144.0f) ? 1u : 0u);
        const int u = missing ? 0 : (int)t - ci;  // |u| < 2^25
        if (kind == 1) return (aux & 3) == 0 ? ((uint32_t)u & 0xffffu) : (uint32_t)(u >> 16);
        const uint64_t w = (uint64_t)((int64_t)u * (int64_t)u) >> (16 * (aux & 3));
        return (aux & 4) ? (uint32_t)w : ((uint32_t)w & 0xffffu);
    }
    // the fractional part in units of 2^-16 m, as an integer: |x - t| < 1, so 3409 of them stay far
    // inside int32, the sums are exact (hence the same for every row block and every kernel that
    // forms them, and a running prefix may be carried from tile to tile), and the quantisation,
    // at most 7.6e-6 m per sample, is two orders below the reference's own float32-FFT floor
    const bool missing = !(fabsf(t) < kMissingLim);  // (a missing sample counts as "no tap" in every sum)
    if (WHAT == kStF) return missing ? 0u : (uint32_t)(int)rintf((x - t) * 65536.0f);
    const int u = missing ? 0 : (int)t - ci;
    if (WHAT == kStU) return (uint32_t)u;
    return (uint32_t)u * (uint32_t)u;
}

// All waves: load the tile's rows, transform, and leave the column prefix sums of it in Q.
// Padded (out-of-domain) samples are staged as 0; the caller accounts for them through m.
// ABS_CLASS (the builds that compute TPI alone): a tile leaves the integer path when a sample is
// non-finite or |trunc(x)| > kAbsLim, whatever c is - int32 sums of 3409 such u cannot wrap, u^2 is
// not needed, and a criterion that does not look at c can be evaluated by the marching build on
// the rows it adds to a carried window.
constexpr float kAbsLim = 262144.0f;
// The scaled route stages (int)rint(x unit): beyond 2^30 / unit the conversion saturates, and a window made of such samples
// alone would have range 0 and pass the unwrapping test (a band of 65535 between the lattice rows of a raster
// whose class said unit = 2^16).  A sample is the scaled chain's only below this limit (2^18 for units up to 2^12, 2^14 at
// 2^16): the others are "bad" like non-finite ones, whatever class the raster was declared with.
__device__ __forceinline__ float scaled_abs_lim(float unit) { return fminf(kAbsLim, 1073741824.0f / unit); }

// Block-wide minimum / maximum of per-lane values through two LDS words (flag_word[2], flag_word[3]; the caller keeps
// them at INT_MAX / INT_MIN between tiles): one butterfly per wave, one atomic pair per wave.
__device__ __forceinline__ void block_range(int* flag_word, int lo, int hi) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        lo = min(lo, __shfl_xor(lo, m));
        hi = max(hi, __shfl_xor(hi, m));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&flag_word[2], lo);
        atomicMax(&flag_word[3], hi);
    }
}

// range (kStS only): smallest and largest staged value of the window, padded samples (0) included.
template <int SIZE, int TH, int NWAVES, int WHAT, typename T, bool ABS_CLASS = false>
__device__ __forceinline__ int stage_prefix(const WaveArgs& p, uint32_t* lds, int* flag_word, int gy0,
                                            int gx, float c, int ci, float lim32, float limcv, int* range = nullptr, int aux = 0) {
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
    uint32_t umax = 0;  // largest |trunc(x) - c| seen, as float bits (NaN / inf sort above all)
    uint32_t tmax = 0;  // largest |trunc(x)| (the builds with STD: a sample beyond 2^18 sends the tile to the limb path whatever c is)
    bool frac = false, any_missing = false;
    int smin = 0x7fffffff, smax = -0x7fffffff - 1;
    Vec4<T> run{{(T)0, (T)0, (T)0, (T)0}};
    if (wave == 0) *reinterpret_cast<Vec4<T>*>(Q + lane * NC) = run;
#pragma unroll
    for (int k = 0; k < SL; ++k) {
        const int r = wave * SL + k;
        if (r < NROWS) {
            const bool ok = row4_inside(p, gy0 + r, gx);
            const bool padded = gy0 + r < 0 || gy0 + r >= p.gny || gx < 0 || gx >= p.nx;  // a tap outside the DEM: the zero padding
#pragma unroll
            for (int s = 0; s < NC; ++s) {
                const float x = v[k].v[s];
                uint32_t bits;
                if (WHAT == kStU) {
                    // one pass classifies the tile and yields u: d = trunc(x) - c is exact in float
                    const float t = truncf(x);
                    const float d = t - c;
                    // (the marching builds - TAKE / ABS_CLASS with p.defer - leave a tile with a missing sample to the general
                    // kernel through kTileFloat, as before; the general kernel tells the two apart)
                    const bool missing = ok && !(fabsf(t) < kMissingLim);
                    const bool counts = ok && !missing;
                    any_missing |= missing;
                    frac |= counts && (x != t);
                    umax = max(umax, counts ? (__float_as_uint(ABS_CLASS ? t : d) & 0x7fffffffu) : 0u);
                    if (!ABS_CLASS) tmax = max(tmax, counts ? (__float_as_uint(t) & 0x7fffffffu) : 0u);
                    bits = counts ? (uint32_t)(int)d : 0u;
                } else {
                    bits = ok ? stage_value<WHAT>(x, c, ci, aux) : 0u;
                    if (WHAT == kStS && ABS_CLASS) {  // the take-all scaled build classifies the window itself
                        const float t = truncf(x);
                        frac |= ok && (x != t);
                        umax = max(umax, ok ? (__float_as_uint(t) & 0x7fffffffu) : 0u);
                    }
                    if (WHAT == kStS && (ok || padded)) {  // (rows of the DEM outside the block view feed no requested output)
                        smin = min(smin, (int)bits);
                        smax = max(smax, (int)bits);
                    }
                }
                T val;
                __builtin_memcpy(&val, &bits, sizeof(T));
                run.v[s] += val;
            }
            *reinterpret_cast<Vec4<T>*>(Q + (r + 1) * ROWW + lane * NC) = run;
        }
    }
    constexpr bool kClassify = WHAT == kStU || (WHAT == kStS && ABS_CLASS);
    if (kClassify) {
        if (frac) flags |= kTileFrac;
        if (!ABS_CLASS && umax > __float_as_uint(lim32)) flags |= kTileWide;
        if (umax > __float_as_uint(ABS_CLASS ? (WHAT == kStS ? scaled_abs_lim(c) : kAbsLim) : limcv) || tmax > __float_as_uint(kAbsLim)) flags |= kTileFloat;
        if (any_missing) flags |= kTileMissing;
    }
    *reinterpret_cast<Vec4<T>*>(TOT + wave * ROWW + lane * NC) = run;
    // tile flags: one ballot per bit inside the wave, one LDS atomic per wave, and the barrier the
    // fix-up pass needs anyway (the flag word is cleared again at the end of the tile)
    if (kClassify) {
        int wf = 0;
        if (__builtin_amdgcn_ballot_w64(flags & kTileFrac)) wf |= kTileFrac;
        if (__builtin_amdgcn_ballot_w64(flags & kTileWide)) wf |= kTileWide;
        if (__builtin_amdgcn_ballot_w64(flags & kTileFloat)) wf |= kTileFloat;
        if (__builtin_amdgcn_ballot_w64(flags & kTileMissing)) wf |= kTileMissing;
        if (lane == 0 && wf) atomicOr(flag_word, wf);
    }
    if (WHAT == kStS) block_range(flag_word, smin, smax);
    __syncthreads();
    const int all = kClassify ? *flag_word : 0;
    if (WHAT == kStS && range != nullptr) {
        range[0] = flag_word[2];
        range[1] = flag_word[3];
    }
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

// With p.defer set (TPI alone) the kernel processes only the tiles the marching build
// (tpi_march_kernel, below) marked for it.
template <int SIZE, int TH, int NWAVES, bool WANT_TPI, bool WANT_STD>
__device__ __forceinline__ void disc_wave_kernel_body(const WaveArgs& p, int tiles_x, int tiles_y, const PartRun deal, const int vb0, const int nb) {
    using G = Geo<SIZE>;
    constexpr bool ABS_CLASS = !WANT_STD;
    // pinned, software-pipelined order of the chain's LDS reads: 3-5 % faster from 45 px (STD 67 px
    // 15.3 -> 14.5 ms), 1-3 % slower at 31 px and below, where a disc has few runs to pipeline
    constexpr bool kPipe = SIZE >= 41;
    constexpr int NROWS = TH + SIZE - 1;
    constexpr int RW = TH / NWAVES;  // output rows per wave
    static_assert(TH % NWAVES == 0, "rows must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    unsigned short* PL = reinterpret_cast<unsigned short*>(lds_u + (NROWS + 1 + NWAVES) * ROWW);
    int* flag_word = reinterpret_cast<int*>(PL + ((TH * (SIZE + 1) + 7) & ~7));
    if (threadIdx.x == 0) *flag_word = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ntiles = tiles_x * tiles_y;
    const int vb = (vb0 + deal.shift) % nb;
    const double n = (double)G::T.taps;
    const double inv_n = 1.0 / n;
    const double inv_nm1 = 1.0 / (n - 1.0);
    const double inv_nn1 = 1.0 / (n * (n - 1.0));
    const float lim32 = floorf(sqrtf(4294967295.0f / (float)G::T.taps));
    const float limcv = fminf(46000.0f, floorf(sqrtf(4294967295.0f / (float)SIZE)));

    // the block's tiles are vb, vb + nb, ...; 64 of them are tested per ballot, so the defer map
    // costs one load per 64 tiles
    for (int base = vb; base < ntiles; base += 64 * nb) {
    const int mine = base + lane * nb;
    bool take = mine < ntiles;
    if (p.defer != nullptr) {
        bool marked = false;
        if (take) {
            if (p.map_th == 0) {
                marked = (p.defer[mine] & 1) != 0;  // kTileGeneral or kTileGeneralRows (2 = waiting for tpi_fraction_march_kernel)
            } else {
                // the map belongs to the marching kernels' geometry (same strips, rows of map_th): this
                // tile is taken when a map tile that shares output rows with it is marked.  Rows of
                // unmarked map tiles get recomputed with identical bits.
                // (with a map from another geometry the tile list is read row-major: a round-robin over a
                // strip-major list whose strips are a multiple of the grid long hands every tile of the top DEM
                // edge to block 0 and every tile of the bottom edge to the last block - 179 tiles each at 32768^2,
                // 7 ms of a launch that has 0.8 ms of work)
                const int tx = mine % tiles_x, ty = mine / tiles_x;
                const int r0 = max((p.out_row0 / TH + ty) * TH, p.out_row0);
                const int r1 = min((p.out_row0 / TH + ty) * TH + TH, p.out_row0 + p.out_rows) - 1;
                const int base = p.out_row0 / p.map_th;
                // (the ring build's strips are wider than this kernel's: map_tw)
                const int mx0 = p.map_tw ? tx * G::TILE_W / p.map_tw : tx;
                const int mx1 = p.map_tw ? (min(tx * G::TILE_W + G::TILE_W, p.nx) - 1) / p.map_tw : tx;
                for (int mx = mx0; mx <= mx1; ++mx)
                    for (int my = r0 / p.map_th; my <= r1 / p.map_th; ++my)
                        marked = marked || (p.defer[mx * p.map_tiles_y + (my - base)] & 1) != 0;
            }
        }
        take = marked;
    }
    unsigned long long todo = __builtin_amdgcn_ballot_w64(take);
    while (todo) {
        const int tile = base + __builtin_ctzll(todo) * nb;
        todo &= todo - 1;
        const bool row_major = p.defer != nullptr && p.map_th != 0;
        // the rows to write: all of them, or - a tile the scaled route left in part - the rows it names
        const uint64_t rows_wanted = (p.defer != nullptr && p.map_th == 0 && p.defer[tile] == kTileGeneralRows) ? p.rowmask[tile] : ~0ull;
        const int ox0 = (row_major ? tile % tiles_x : tile / tiles_y) * G::TILE_W;
        const int oy0 = (p.out_row0 / TH + (row_major ? tile / tiles_x : tile % tiles_y)) * TH;  // global multiples of TH
        const int gx = ox0 - G::X0 + lane * NC;
        const int gy0 = oy0 + G::T.off_min;
        int cy = min(max(oy0 + TH / 2, 0), p.gny - 1);
        cy = min(max(cy, p.in_row0), p.in_row0 + p.in_rows - 1);
        const int cx = min(ox0 + G::TILE_W / 2, p.nx - 1);
        float c = truncf(p.in[(size_t)(cy - p.in_row0) * p.nx + cx]);
        if (!(fabsf(c) < kMissingLim)) c = 0.0f;  // (any value would do: the sums are exact integers and the results do not depend on it)
        const int ci = (int)c;
        const bool border = gy0 < 0 || gy0 + NROWS > p.gny || ox0 - G::X0 < 0 || ox0 - G::X0 + ROWW > p.nx;

        // per-row sums live in this block's scratch planes between the passes (registers cannot
        // hold RW x NC x 4 values next to the chain): plane 0 Su (int32), 1/2 Su2 low/high 16-bit half sums, 3 Sf; on a tile
        // of the limb path 0/4 Su (64 bits), 1/2 Su2 (64 bits), 5 the counts of kStInd.  Written and read by the same lane.
        uint32_t* plane = p.scratch + (size_t)blockIdx.x * (kScratchPlanes * TH * ROWW) + lane * NC;
        auto put = [&](int which, int jj, const uint32_t (&val)[NC]) {
            Vec4<uint32_t> x{{val[0], val[1], val[2], val[3]}};
            *reinterpret_cast<Vec4<uint32_t>*>(plane + (which * TH + jj) * ROWW) = x;
        };
        auto get = [&](int which, int jj) {
            return *reinterpret_cast<const Vec4<uint32_t>*>(plane + (which * TH + jj) * ROWW);
        };

        const int flags = stage_prefix<SIZE, TH, NWAVES, kStU, int, ABS_CLASS>(p, lds_u, flag_word, gy0, gx, c, ci, lim32, limcv);
        const bool use_float = (flags & kTileFloat) != 0;      // the limb passes
        const bool has_missing = (flags & kTileMissing) != 0;  // one more pass counts the missing samples per disc
        const bool wide = (flags & kTileWide) != 0;
        const bool frac = (flags & kTileFrac) != 0;
        // TPI alone on an integer-valued tile needs one pass: its rows are finalised straight from
        // the chain, without the round trip through the scratch planes
        const bool direct = !WANT_STD && !use_float && !frac && !has_missing;
        const int ocol = ox0 + lane * NC;
        const bool lane_ok = lane < G::NVL && ocol < p.nx;
        auto finalise_row = [&](int jj, const Vec4<uint32_t>& q0, const Vec4<uint32_t>& q1,
                                const Vec4<uint32_t>& q2, const Vec4<uint32_t>& q3) {
            const int oy = oy0 + jj;
            if (!lane_ok || oy < p.out_row0 || oy >= p.out_row0 + p.out_rows || !((rows_wanted >> jj) & 1)) return;
            Vec4<uint32_t> ind{{0u, 0u, 0u, 0u}};
            if (has_missing) ind = get(5, jj);  // (kStInd: missing samples in the disc, in the upper half)
            Vec4<float> xs{{0.f, 0.f, 0.f, 0.f}};
            if (WANT_TPI) xs = *reinterpret_cast<const Vec4<float>*>(p.in + (size_t)(oy - p.in_row0) * p.nx + ocol);
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
                const double sf = (double)(int)q3.v[t] * (1.0 / 65536.0);  // exact in float64
                // sums of u and u^2 over the in-domain taps
                const double su = (double)(int)q0.v[t];
                const double su2 = (double)q2.v[t] * 65536.0 + (double)q1.v[t];  // q2 = 0 unless the tile is wide
                const double cd = (double)c;
                // su + c m is an exact integer (= sum of trunc(x)), so the result does not depend on
                // which c the tile happened to use
                const double s1 = (su + cd * m) + sf;
                if (WANT_STD) {
                    // (per pixel, not per tile: a window of whole metres inside a tile that has fractional samples
                    // elsewhere must get the bits the ring kernel, whose flags cover a batch of rows, gives it; found
                    // by the randomised row-block check once a run's hand-over depended on the run's extent)
                    if (m == n && q3.v[t] == 0) {
                        out_s.v[t] = std_from_int_sums((int)q0.v[t], ((uint64_t)q2.v[t] << 16) + q1.v[t], (uint32_t)G::T.taps,
                                                       (float)inv_nn1);
                    } else if (q3.v[t] == 0) {  // whole metres, taps outside the DEM
                        out_s.v[t] = std_from_border_sums((int64_t)(int)q0.v[t], ((uint64_t)q2.v[t] << 16) + q1.v[t], ci, (int)m,
                                                          (uint32_t)G::T.taps, (float)inv_nn1);
                    } else {
                        const double s2 = su2 + 2.0 * cd * su + cd * cd * m;
                        out_s.v[t] = std_from_sums(s1, s2, inv_n, inv_nm1);
                    }
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
            if (has_missing) {
#pragma unroll
                for (int t = 0; t < NC; ++t) {
                    if ((ind.v[t] >> 16) != 0) out_s.v[t] = out_t.v[t] = __uint_as_float(0x7fc00000u);  // a missing sample in the disc: no value
                }
            }
            const size_t o = (size_t)(oy - p.out_row0) * p.nx + ocol;
            if (WANT_STD) *reinterpret_cast<Vec4<float>*>(p.sd + o) = out_s;
            if (WANT_TPI) *reinterpret_cast<Vec4<float>*>(p.tpi + o) = out_t;
        };
        // a row of a tile of the limb path, from the scratch planes
        auto finalise_limb_row = [&](int jj) {
            const int oy = oy0 + jj;
            if (!lane_ok || oy < p.out_row0 || oy >= p.out_row0 + p.out_rows || !((rows_wanted >> jj) & 1)) return;
            Vec4<float> xs{{0.f, 0.f, 0.f, 0.f}};
            if (WANT_TPI) xs = *reinterpret_cast<const Vec4<float>*>(p.in + (size_t)(oy - p.in_row0) * p.nx + ocol);
            const Vec4<uint32_t> ul = get(0, jj), uh = get(4, jj), ind = get(5, jj);
            Vec4<uint32_t> wl{{0u, 0u, 0u, 0u}}, wh{{0u, 0u, 0u, 0u}}, q3{{0u, 0u, 0u, 0u}};
            if (WANT_STD) {
                wl = get(1, jj);
                wh = get(2, jj);
            }
            if (frac) q3 = get(3, jj);
            Vec4<float> out_t, out_s;
#pragma unroll
            for (int t = 0; t < NC; ++t) {
                int mi = G::T.taps;
                if (border) {
                    const int d_lo = max(G::T.off_min, -(ocol + t));
                    const int d_hi = min(G::T.off_max, p.nx - 1 - (ocol + t));
                    const unsigned short* pl = PL + jj * (SIZE + 1) - G::T.off_min;
                    mi = d_hi >= d_lo ? (int)pl[d_hi + 1] - (int)pl[d_lo] : 0;
                }
                const int64_t Su = (int64_t)(((uint64_t)uh.v[t] << 32) | ul.v[t]);
                const uint64_t Su2 = ((uint64_t)wh.v[t] << 32) | wl.v[t];
                const int sg = (int)q3.v[t];
                const int64_t T = Su + (int64_t)ci * mi;  // sum of trunc(x): whatever c was
                const __int128 S2 = (__int128)Su2 + 2 * (__int128)ci * Su + (__int128)((int64_t)ci * ci) * mi;  // sum of trunc(x)^2
                const double sf = (double)sg * (1.0 / 65536.0);
                const double s1 = (double)T + sf;
                const bool missing = (ind.v[t] >> 16) != 0, ordinary = (ind.v[t] & 0xffffu) == 0;
                float sd_v, tpi_v = 0.0f;
                if (ordinary) {
                    // a disc of ordinary samples: the expressions of every other kernel on the same exact sums
                    if (sg == 0) {
                        // n S2 - T^2 (= n Su2 - Su^2 when the disc is whole: std_from_int_sums / std_from_border_sums), below 2^60 here
                        const uint64_t num = (uint64_t)((__int128)G::T.taps * S2 - (__int128)T * T);
                        const float f = fmaf((float)(uint32_t)(num >> 32), 4294967296.0f, (float)(uint32_t)num);
                        sd_v = __builtin_amdgcn_sqrtf(f * (float)inv_nn1);
                    } else {
                        sd_v = std_from_sums(s1, (double)(int64_t)S2, inv_n, inv_nm1);
                    }
                } else {
                    sd_v = std_from_exact_sums(T, S2, sg, G::T.taps, inv_nn1);
                }
                if (WANT_TPI) {
                    const int cy2 = oy + G::T.centre, cx2 = ocol + t + G::T.centre;
                    double x_ctr = (double)xs.v[t];
                    if (G::T.centre != 0) {
                        const bool in = cy2 >= 0 && cy2 < p.gny && cx2 >= 0 && cx2 < p.nx;
                        x_ctr = in ? (double)p.in[(size_t)(cy2 - p.in_row0) * p.nx + cx2] : 0.0;
                    }
                    tpi_v = (float)((double)xs.v[t] - (s1 - x_ctr) * inv_nm1);
                }
                // a missing sample in the disc (non-finite, or beyond +-2^24): no value
                out_s.v[t] = missing ? __uint_as_float(0x7fc00000u) : sd_v;
                out_t.v[t] = missing ? __uint_as_float(0x7fc00000u) : tpi_v;
            }
            const size_t o = (size_t)(oy - p.out_row0) * p.nx + ocol;
            if (WANT_STD) *reinterpret_cast<Vec4<float>*>(p.sd + o) = out_s;
            if (WANT_TPI) *reinterpret_cast<Vec4<float>*>(p.tpi + o) = out_t;
        };
        if (border && threadIdx.x < TH) {
            // in-domain rows of each column run, prefix-summed over the column offsets: what the
            // zero-padded convolution really sums over near the DEM border
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
        if (direct && border) __syncthreads();  // the border table is read right away
        if (!use_float) {
#pragma unroll 1
            for (int k = 0; k < RW; ++k) {
                int acc[NC];
                wave_disc_sum<SIZE, int, 0, kPipe>(reinterpret_cast<const int*>(lds_u), wave + k * NWAVES, lane, acc);
                const uint32_t bits[NC] = {(uint32_t)acc[0], (uint32_t)acc[1], (uint32_t)acc[2], (uint32_t)acc[3]};
                if (direct) {
                    const Vec4<uint32_t> q0{{bits[0], bits[1], bits[2], bits[3]}}, z{{0u, 0u, 0u, 0u}};
                    finalise_row(wave + k * NWAVES, q0, z, z, z);
                } else {
                    put(0, wave + k * NWAVES, bits);
                }
            }
            if (WANT_STD) {
            __syncthreads();
            stage_prefix<SIZE, TH, NWAVES, kStU2, uint32_t>(p, lds_u, flag_word, gy0, gx, c, ci, lim32, limcv);
            if (!wide) {
#pragma unroll 1
                for (int k = 0; k < RW; ++k) {
                    uint32_t acc[NC];
                    wave_disc_sum<SIZE, uint32_t, 0, kPipe>(lds_u, wave + k * NWAVES, lane, acc);
                    put(1, wave + k * NWAVES, acc);
                }
            } else {
#pragma unroll 1
                for (int k = 0; k < RW; ++k) {
                    uint32_t lo[NC], hi[NC];  // sums of the low / high 16 bits of the column sums
                    wave_disc_sum<SIZE, uint32_t, 1, kPipe>(lds_u, wave + k * NWAVES, lane, lo);
                    wave_disc_sum<SIZE, uint32_t, 2, kPipe>(lds_u, wave + k * NWAVES, lane, hi);
                    put(1, wave + k * NWAVES, lo);
                    put(2, wave + k * NWAVES, hi);
                }
            }
            }
            if (has_missing) {
                // missing samples (staged as u = 0 by the passes above): their count per disc, so that exactly the pixels
                // whose discs hold one come out as NaN
                __syncthreads();
                stage_prefix<SIZE, TH, NWAVES, kStLimb, uint32_t>(p, lds_u, flag_word, gy0, gx, c, ci, lim32, limcv, nullptr, 0);
#pragma unroll 1
                for (int k = 0; k < RW; ++k) {
                    uint32_t acc[NC];
                    wave_disc_sum<SIZE, uint32_t, 0, kPipe>(lds_u, wave + k * NWAVES, lane, acc);
                    put(5, wave + k * NWAVES, acc);
                }
            }
        } else {
            // the limb path (Stage, above): counts of the samples that are not ordinary / missing, then u and u^2 limb by limb
            // one copy of the staging and chain code, six passes: {aux (kind << 4 | limb | last << 2), planes, shift, signed, first}
            const int npass = WANT_STD ? 6 : 3;
#pragma unroll 1
            for (int pass = 0; pass < npass; ++pass) {
                // pass:      0 ind   1 u lo   2 u hi   3 u^2 [0,16)   4 u^2 [16,32)   5 u^2 [32, ...)
                const int aux = pass == 0 ? 0 : (pass <= 2 ? (1 << 4) | (pass - 1) : (2 << 4) | (pass - 3) | (pass == 5 ? 4 : 0));
                const int plo = pass == 0 ? 5 : (pass <= 2 ? 0 : 1), phi = pass == 0 ? -1 : (pass <= 2 ? 4 : 2);
                const int shift = pass == 0 ? 0 : (pass <= 2 ? 16 * (pass - 1) : 16 * (pass - 3));
                const bool is_signed = pass == 2, first = pass == 1 || pass == 3;
                __syncthreads();
                stage_prefix<SIZE, TH, NWAVES, kStLimb, uint32_t>(p, lds_u, flag_word, gy0, gx, c, ci, lim32, limcv, nullptr, aux);
#pragma unroll 1
                for (int k = 0; k < RW; ++k) {
                    const int jj = wave + k * NWAVES;
                    uint32_t acc[NC];  // the limb's sum over the disc: below 2^32 (as a signed number for the upper limb of u)
                    wave_disc_sum<SIZE, uint32_t, 0, kPipe>(lds_u, jj, lane, acc);
                    if (phi < 0) {
                        put(plo, jj, acc);
                        continue;
                    }
                    Vec4<uint32_t> lo{{0u, 0u, 0u, 0u}}, hi{{0u, 0u, 0u, 0u}};
                    if (!first) {
                        lo = get(plo, jj);
                        hi = get(phi, jj);
                    }
                    uint32_t nlo[NC], nhi[NC];
#pragma unroll
                    for (int t = 0; t < NC; ++t) {
                        const uint64_t add = (is_signed ? (uint64_t)(int64_t)(int)acc[t] : (uint64_t)acc[t]) << shift;
                        const uint64_t v = (((uint64_t)hi.v[t] << 32) | lo.v[t]) + add;
                        nlo[t] = (uint32_t)v;
                        nhi[t] = (uint32_t)(v >> 32);
                    }
                    put(plo, jj, nlo);
                    put(phi, jj, nhi);
                }
            }
        }
        if (frac) {
            __syncthreads();
            stage_prefix<SIZE, TH, NWAVES, kStF, uint32_t>(p, lds_u, flag_word, gy0, gx, c, ci, lim32, limcv);
#pragma unroll 1
            for (int k = 0; k < RW; ++k) {
                uint32_t acc[NC];  // sum of the fractional parts in units of 2^-16 m, modulo 2^32 (fits int32)
                wave_disc_sum<SIZE, uint32_t, 0, kPipe>(lds_u, wave + k * NWAVES, lane, acc);
                put(3, wave + k * NWAVES, acc);
            }
        }

        // ---- finalise the leftover rows from the scratch planes ------------------------------------
        if (use_float) {
#pragma unroll 1
            for (int k = 0; k < RW; ++k) finalise_limb_row(wave + k * NWAVES);
        } else if (!direct) {
#pragma unroll 1
            for (int k = 0; k < RW; ++k) {
                const int jj = wave + k * NWAVES;
                const Vec4<uint32_t> q0 = get(0, jj);
                Vec4<uint32_t> q1{{0u, 0u, 0u, 0u}}, q2{{0u, 0u, 0u, 0u}}, q3{{0u, 0u, 0u, 0u}};
                if (WANT_STD) q1 = get(1, jj);
                if (WANT_STD && wide) q2 = get(2, jj);
                if (frac) q3 = get(3, jj);
                finalise_row(jj, q0, q1, q2, q3);
            }
        }
        if (threadIdx.x == 0) *flag_word = 0;
        __syncthreads();  // Q, PL and the flag word are rewritten by the next tile
    }
    }
}

template <int SIZE, int TH, int NWAVES, bool WANT_TPI, bool WANT_STD>
__global__ __launch_bounds__(NWAVES * 64) void disc_wave_kernel(WaveParts ps, int tiles_x) {
    TOPO_RUN_PARTS_LOOP((disc_wave_kernel_body<SIZE, TH, NWAVES, WANT_TPI, WANT_STD>));
}

// ---- TPI alone on tiles of whole metres: the marching build --------------------------------------
// The general kernel stages TH + SIZE - 1 rows to produce TH of them; for 67 px that is 126 rows for
// 60, and the 66 extra rows are the ones the tile above staged a moment ago.  This build gives each
// persistent block a contiguous run of tiles going down a column strip and keeps them: the last
// SIZE prefix rows of a tile are moved to the top of the LDS image (LDS -> LDS, 67 KiB), the prefix
// sums of the TH new rows continue from the last carried row, and only those rows are loaded and
// classified.  It can do so because everything on this path is an integer:
//   * u = trunc(x) with no offset (c = 0): int32 sums of 3409 samples of |x| <= kAbsLim cannot wrap,
//     and a running int32 prefix may wrap freely - only differences over <= SIZE rows are used;
//   * padded taps are staged as 0 and so drop out of sum(x): no tap counting near the DEM border;
//   * on such a tile x = trunc(x), so the pixel's own value is the difference of two prefix rows
//     and is not re-read from the DEM.
// The finalisation evaluates the same float64 expression as the general kernel's integer path on
// the same exact integer sum(x), so both give the same bits.  A tile with a fractional, non-finite
// or absurd sample is marked in p.defer and left to the general kernel (launched right after with
// only_deferred); the carried window is dropped and the next tile staged in full.  After kGiveUp
// such tiles in a row the block marks the rest of its run without staging it.
// FRACTION: stage the fractional parts (integers in units of 2^-16 m, stage_value<kStF>) instead of
// trunc(x), without classifying: the window was classified when its trunc(x) sums were formed.
// -DMARCH_STAMPS: s_memtime stamps around the phases of a marched tile, summed per wave and printed by two blocks
// (diagnostic build of tools/ubench/tpi_lab.hip; the product is built without it)
#ifdef MARCH_STAMPS
#define MARCH_STAMP(i) { const long long now_ = __builtin_amdgcn_s_memtime(); tsum[i] += now_ - tlast; tlast = now_; }
#define MARCH_STAMP_ARGS , long long (&tsum)[8], long long& tlast
#define MARCH_STAMP_PASS , tsum, tlast
#else
#define MARCH_STAMP(i)
#define MARCH_STAMP_ARGS
#define MARCH_STAMP_PASS
#endif

// MARCH_DYN_ROWS (default on): the waves of a block draw the output rows of a tile from a ticket counter in LDS
// (flag_word[1]) instead of owning every NWAVES-th row, so a wave that the SIMD's arbitration held back takes
// fewer rows and the tile's row loop ends for all waves within a row's time.  Which wave computes a row does not
// change its bits.
#ifndef MARCH_DYN_ROWS
#define MARCH_DYN_ROWS 1
#endif

// MODE 0: trunc(x), classified; 1 (the former FRACTION): kStF; 2: kStS, with the range of the NEW rows in range[0 .. 1];
// 3: kStS, range and classification (the take-all scaled build)
template <int SIZE, int TH, int NWAVES, int MODE = 0>
__device__ __forceinline__ int stage_march(const WaveArgs& p, uint32_t* Q, int* flag_word, int gy0, int gx MARCH_STAMP_ARGS,
                                           int* range = nullptr) {
    constexpr bool FRACTION = MODE == 1 || MODE == 2;  // (no classification)
    constexpr bool SCALED = MODE == 2 || MODE == 3;
    constexpr int NROWS = TH + SIZE - 1;
    constexpr int KEEP = NROWS + 1 - TH;  // prefix rows carried over: those of window rows TH-1 .. NROWS-1
    constexpr int RW = TH / NWAVES;       // new rows per wave
    constexpr int NT = NWAVES * 64;
    constexpr int MOVE4 = KEEP * 64;      // 16-byte pieces to move
    constexpr int MOVES = (MOVE4 + NT - 1) / NT;
    uint32_t* TOT = Q + (NROWS + 1) * ROWW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r0 = KEEP - 1 + wave * RW;  // first new window row of this wave
    Vec4<float> v[RW];
#pragma unroll
    for (int k = 0; k < RW; ++k) v[k] = load_row4(p, gy0 + r0 + k, gx);
    // old prefix rows TH .. NROWS become rows 0 .. KEEP-1: read, barrier (also: every wave is done
    // reading the previous tile's image), write
    typedef uint32_t int4v __attribute__((ext_vector_type(4)));  // a native vector: plain 16-byte loads / stores
    int4v m[MOVES];
    const int4v* src = reinterpret_cast<const int4v*>(Q + TH * ROWW);
#pragma unroll
    for (int s = 0; s < MOVES; ++s) {
        // unconditional (clamped) so that m[] stays in registers; the write below is guarded
        const int idx = (int)threadIdx.x + s * NT;
        m[s] = src[idx < MOVE4 ? idx : MOVE4 - 1];
    }
    MARCH_STAMP(0)
    __syncthreads();
    MARCH_STAMP(1)
    if (MARCH_DYN_ROWS && threadIdx.x == 0) flag_word[1] = 0;  // every wave has left the previous tile's row loop
    int4v* dst = reinterpret_cast<int4v*>(Q);
#pragma unroll
    for (int s = 0; s < MOVES; ++s) {
        const int idx = (int)threadIdx.x + s * NT;
        if (idx < MOVE4) dst[idx] = m[s];
    }
    uint32_t amax = 0;
    bool frac = false;
    int smin = 0x7fffffff, smax = -0x7fffffff - 1;
    Vec4<uint32_t> run{{0u, 0u, 0u, 0u}};  // unsigned: the running prefix is allowed to wrap
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        const bool ok = row4_inside(p, gy0 + r0 + k, gx);
        const bool padded = gy0 + r0 + k < 0 || gy0 + r0 + k >= p.gny || gx < 0 || gx >= p.nx;
#pragma unroll
        for (int s = 0; s < NC; ++s) {
            const float x = v[k].v[s];
            if (SCALED) {
                const uint32_t q = ok ? stage_value<kStS>(x, p.unit, 0) : 0u;
                if (ok || padded) {
                    smin = min(smin, (int)q);
                    smax = max(smax, (int)q);
                }
                if (MODE == 3) {
                    const float t = truncf(x);
                    frac |= ok && (x != t);
                    amax = max(amax, ok ? (__float_as_uint(t) & 0x7fffffffu) : 0u);
                }
                run.v[s] += q;
            } else if (MODE == 1) {
                run.v[s] += ok ? stage_value<kStF>(x, 0.0f, 0) : 0u;
            } else {
                const float t = truncf(x);
                frac |= ok && (x != t);
                amax = max(amax, ok ? (__float_as_uint(t) & 0x7fffffffu) : 0u);
                run.v[s] += ok ? (uint32_t)(int)t : 0u;
            }
        }
        *reinterpret_cast<Vec4<uint32_t>*>(Q + (r0 + k + 1) * ROWW + lane * NC) = run;
    }
    *reinterpret_cast<Vec4<uint32_t>*>(TOT + wave * ROWW + lane * NC) = run;
    if (!FRACTION) {
        int wf = 0;
        if (__builtin_amdgcn_ballot_w64(frac)) wf |= kTileFrac;
        if (__builtin_amdgcn_ballot_w64(amax > __float_as_uint(SCALED ? scaled_abs_lim(p.unit) : kAbsLim))) wf |= kTileFloat;
        if (lane == 0 && wf) atomicOr(flag_word, wf);
    }
    if (SCALED) block_range(flag_word, smin, smax);
    MARCH_STAMP(2)
    __syncthreads();
    MARCH_STAMP(3)
    const int all = FRACTION ? 0 : *flag_word;
    if (SCALED && range != nullptr) {
        range[0] = flag_word[2];
        range[1] = flag_word[3];
    }
    // every wave adds the carried prefix (row KEEP-1) and the totals of the waves above it
    Vec4<uint32_t> off = *reinterpret_cast<const Vec4<uint32_t>*>(Q + (KEEP - 1) * ROWW + lane * NC);
    for (int w = 0; w < wave; ++w) {
        const Vec4<uint32_t> t = *reinterpret_cast<const Vec4<uint32_t>*>(TOT + w * ROWW + lane * NC);
#pragma unroll
        for (int s = 0; s < NC; ++s) off.v[s] += t.v[s];
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        Vec4<uint32_t>* q = reinterpret_cast<Vec4<uint32_t>*>(Q + (r0 + k + 1) * ROWW + lane * NC);
        Vec4<uint32_t> x = *q;
#pragma unroll
        for (int s = 0; s < NC; ++s) x.v[s] += off.v[s];
        *q = x;
    }
    MARCH_STAMP(4)
    __syncthreads();
    MARCH_STAMP(5)
    return all;
}

// OUT_TPI: write TPI.  OUT_SUM: write the exact disc sums of trunc(x) to p.sums, for the STD kernel
// that follows (std_march_kernel).  ALLOW_FRAC (TPI alone): a tile with fractional samples is not
// left to the general kernel; its sums of trunc(x), which are exact whatever the fractional parts
// are, go to p.sums instead of TPI, and the tile is marked kNeedsFraction for
// tpi_fraction_march_kernel, which adds the sum of the fractional parts and finalises.

// SUMS_ALL (with ALLOW_FRAC): the sums go to p.sums for EVERY tile - the first of the three marching passes of STD on a DEM with
// fractional elevations (launch_wave_any): std_march_kernel needs them for the whole-metre tiles, the fraction pass for the others.
template <int SIZE, int TH, int NWAVES, bool OUT_TPI, bool OUT_SUM, bool ALLOW_FRAC = false, bool SUMS_ALL = false>
__device__ __forceinline__ void tpi_march_kernel_body(const WaveArgs& p, int tiles_x, int tiles_y, const PartRun deal, const int vb0, const int nb) {
    using G = Geo<SIZE>;
    static_assert(OUT_TPI || OUT_SUM, "nothing to write");
    static_assert(G::T.centre == 0, "odd disc sizes only: the zeroed tap is the pixel itself");
    static_assert(TH % NWAVES == 0, "rows must split evenly over the waves");
    constexpr int NROWS = TH + SIZE - 1;
    constexpr int kGiveUp = 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    uint32_t* Q = lds_u;  // uint32 prefix sums of trunc(x): wrap-around is defined and harmless
    int* flag_word = reinterpret_cast<int*>(Q + (NROWS + 1 + NWAVES) * ROWW);
    if (threadIdx.x == 0) *flag_word = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ntiles = tiles_x * tiles_y;
    const int vb = (vb0 + deal.shift) % nb;
    // a contiguous run of the strip-major tile list per block; neighbouring runs sit in one XCD
    // the block's contiguous run of the strip-major tile list (deal_parts)
    const int first = deal.first(vb);
    const int last = min(first + deal.count(vb), ntiles);
    const double inv_nm1 = 1.0 / ((double)G::T.taps - 1.0);

    bool carry = false;  // the LDS image holds the window of the tile right above
    // ALLOW_FRAC keeps the carry across tiles with fractional samples, and a marched staging pass
    // classifies only the TH rows it adds.  The SIZE - 1 carried rows are the new rows of the last
    // kHist tiles, so "fractional" is kept as a history of those tiles' flags: bit 0 the current
    // tile's new rows, bit k those of the k-th tile above.  (Marking a window fractional that no
    // longer holds a fractional sample only costs time: its fractional sum is exactly 0 and the
    // result has the same bits.)  Without this, e.g. the last tile of a strip, whose new rows lie
    // below the DEM, would drop the fractional parts of the rows it carries.
    constexpr int kHist = (SIZE - 1 + TH - 1) / TH;
    unsigned frac_hist = 0;
    int deferred_in_a_row = 0;
    int seen_frac = 0;  // tiles of the run with fractional samples in their window (the report to dem_memo)
#ifdef MARCH_STAMPS
    long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (int tile = first; tile < last; ++tile) {
        // (the scaled route decides row by row what it can take of a tile with a non-finite or absurd sample: its pass gets
        // the tile, not the general kernel)
        const uint8_t left_to = ALLOW_FRAC && p.scaled != 0 ? kNeedsFraction : kTileGeneral;
        if (deferred_in_a_row >= kGiveUp) {
            if (threadIdx.x == 0) p.defer[tile] = left_to;
            continue;
        }
        const int ty = tile % tiles_y;
        const int ox0 = (tile / tiles_y) * G::TILE_W;
        const int oy0 = (p.out_row0 / TH + ty) * TH;  // global multiples of TH
        const int gx = ox0 - G::X0 + lane * NC;
        const int gy0 = oy0 + G::T.off_min;
        if (ty == 0) carry = false;  // top of a strip
        int flags;
        if (carry) {
            MARCH_STAMP(7)
            flags = stage_march<SIZE, TH, NWAVES>(p, Q, flag_word, gy0, gx MARCH_STAMP_PASS);
        } else {
            __syncthreads();  // the image and the flag word of the previous tile are done with
            if (MARCH_DYN_ROWS && threadIdx.x == 0) flag_word[1] = 0;
            flags = stage_prefix<SIZE, TH, NWAVES, kStU, int, true>(p, lds_u, flag_word, gy0, gx, 0.0f, 0, 0.0f, 0.0f);
#ifdef MARCH_STAMPS
            tlast = __builtin_amdgcn_s_memtime();  // full stagings (run starts) are not in the sums
#endif
        }
        const bool leave = (flags & (ALLOW_FRAC ? (kTileFloat | kTileMissing) : (kTileFloat | kTileMissing | kTileFrac))) != 0;
        if (ALLOW_FRAC) {
            const unsigned now = (flags & kTileFrac) ? 1u : 0u;
            // a full staging pass classified the whole window: it stands for every row carried on
            frac_hist = carry ? (((frac_hist << 1) | now) & ((2u << kHist) - 1u)) : (now ? (2u << kHist) - 1u : 0u);
        }
        const bool fraction = ALLOW_FRAC && !leave && frac_hist != 0;  // sums only, TPI later
        if (threadIdx.x == 0) {
            p.defer[tile] = leave ? left_to : (fraction ? kNeedsFraction : kTileDone);
            *flag_word = 0;  // every thread has read it; the next atomicOr is behind a barrier
        }
        if (leave) {
            ++deferred_in_a_row;
            carry = false;
            continue;
        }
        deferred_in_a_row = 0;
        carry = true;

        const int ocol = ox0 + lane * NC;
        const bool lane_ok = lane < G::NVL && ocol < p.nx;
        // one copy of the row loop per kind of tile (the choice is the same for the whole block), so
        // that the whole-metre copy is exactly the loop of the kernel without ALLOW_FRAC
        auto rows = [&](auto fraction_tag) {
            constexpr bool SUMS_ONLY = decltype(fraction_tag)::value;  // fractional tile: sums now, TPI later
            // the next row of the tile that nobody has taken yet (MARCH_DYN_ROWS), drawn one row ahead so that the
            // LDS atomic's round trip is behind the chain
            auto draw = [&]() {
                int t = 0;
                if (lane == 0) t = atomicAdd(&flag_word[1], 1);
                return __builtin_amdgcn_readfirstlane(t);
            };
            int next_row = MARCH_DYN_ROWS ? draw() : wave;
#pragma unroll 1
            while (next_row < TH) {
                const int jj = next_row;
                next_row = MARCH_DYN_ROWS ? draw() : jj + NWAVES;
                const int oy = oy0 + jj;
                // (a row outside the output rows - the first and last tile of a row block, and most of a seam tile of a
                // row shard - costs no chain; the test is on scalars)
                if (oy < p.out_row0 || oy >= p.out_row0 + p.out_rows) continue;
                uint32_t acc[NC];  // sum of trunc(x) over the disc modulo 2^32; the true value fits int32
                wave_disc_sum<SIZE, uint32_t, 0, (SIZE >= 41)>(Q, jj, lane, acc);
                if (!lane_ok) continue;
                const size_t o = (size_t)(oy - p.out_row0) * p.nx + ocol;
                if (OUT_SUM && (!ALLOW_FRAC || SUMS_ONLY || SUMS_ALL)) {
                    const Vec4<int> sv{{(int)acc[0], (int)acc[1], (int)acc[2], (int)acc[3]}};
                    *reinterpret_cast<Vec4<int>*>(p.sums + o) = sv;
                }
                if (OUT_TPI && !SUMS_ONLY) {
                    // the pixel's own (integer) value: prefix through its row minus prefix above it
                    const uint32_t* own = Q + (jj - G::T.off_min) * ROWW + lane * NC + G::X0;
                    const Vec4<uint32_t> hi = *reinterpret_cast<const Vec4<uint32_t>*>(own + ROWW);
                    const Vec4<uint32_t> lo = *reinterpret_cast<const Vec4<uint32_t>*>(own);
                    Vec4<float> out_t;
#pragma unroll
                    for (int t = 0; t < NC; ++t) {
                        // x and (sum of x over the in-domain taps) - x are integers: the difference is taken in int32
                        // and each goes to float64 with one conversion (the values, hence the bits, of the form
                        // (double)(float)x - ((double)s1 - (double)(float)x) * inv_nm1, two conversions less per pixel)
                        const int xi = (int)(hi.v[t] - lo.v[t]);
                        out_t.v[t] = (float)((double)xi - (double)((int)acc[t] - xi) * inv_nm1);
                    }
                    *reinterpret_cast<Vec4<float>*>(p.tpi + o) = out_t;
                }
            }
        };
        if (ALLOW_FRAC && fraction) {
            ++seen_frac;
            if (p.scaled == 0) rows(std::true_type{});  // (scaled route: tpi_scaled_march_kernel needs no sums)
        } else {
            rows(std::false_type{});
        }
        MARCH_STAMP(6)
    }
    if (ALLOW_FRAC && p.report != nullptr && vb == nb / 2 && threadIdx.x == 0) {
        __hip_atomic_store(p.report + 1, (uint32_t)seen_frac, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(p.report, (uint32_t)(last > first ? last - first : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#ifdef MARCH_STAMPS
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 131) && (wave == 0 || wave == 5 || wave == 11))
        printf("blk %3d wave %2d tiles %d: loads+move reads %lld | barrier %lld | move writes+scan+write %lld | barrier %lld | "
               "fix-up %lld | barrier %lld | rows (chain+finalise+store) %lld | tile set-up %lld   (memtime ticks, 100 MHz)\n",
               (int)blockIdx.x, wave, last - first, tsum[0], tsum[1], tsum[2], tsum[3], tsum[4], tsum[5], tsum[6], tsum[7]);
#endif
}

template <int SIZE, int TH, int NWAVES, bool OUT_TPI, bool OUT_SUM, bool ALLOW_FRAC = false>
__global__ __launch_bounds__(NWAVES * 64) void tpi_march_kernel(WaveArgs p, int tiles_x, int tiles_y, PartRun deal) {
    TOPO_RUN_ONE((tpi_march_kernel_body<SIZE, TH, NWAVES, OUT_TPI, OUT_SUM, ALLOW_FRAC>));
}
template <int SIZE, int TH, int NWAVES, bool OUT_TPI, bool OUT_SUM, bool ALLOW_FRAC = false>
__global__ __launch_bounds__(NWAVES * 64) void tpi_march_kernel_parts(WaveParts ps, int tiles_x) {
    TOPO_RUN_PARTS((tpi_march_kernel_body<SIZE, TH, NWAVES, OUT_TPI, OUT_SUM, ALLOW_FRAC>));
}
template <int SIZE, int TH, int NWAVES, bool OUT_TPI>
__global__ __launch_bounds__(NWAVES * 64) void tpi_march_sums_kernel(WaveParts ps, int tiles_x) {  // (SUMS_ALL: one form for every call)
    TOPO_RUN_PARTS_LOOP((tpi_march_kernel_body<SIZE, TH, NWAVES, OUT_TPI, true, true, true>));
}

// Grid of the marching launches: persistent blocks, whole XCD rounds, never more blocks than tiles.
// While a ghost-row exchange is in flight the grid is one round of 8 shorter per reserved CU-round: a kernel's
// workgroups go to the 8 XCDs round-robin, so a grid of whole rounds leaves the same number of CUs free on EVERY XCD,
// which is where RCCL's 8 workgroups (one per XCD, cap_rccl_channels) land.  Measured with tools/ubench/xcd_map.hip:
// next to 248 resident blocks a second kernel's 8 workgroups start at once; next to 252 only the one whose XCD has the
// free CU does; and a CU whose workgroup has left early is NOT handed to another kernel while the launch is resident.
inline long march_grid(Context& c, int blocks_per_cu, long ntiles) {
    const int rounds_off = (c.reserve_cus + 7) / 8;
    long grid = (long)(c.num_cu - 8 * rounds_off) * blocks_per_cu;
    if (grid < 8) grid = 8;
    grid -= grid % 8;
    return grid > ntiles ? ntiles : grid;
}

// Launches `kernel` over the parts and, when this launch took the gate of a sharded call, the clean-up launch
// behind the exchange's event: an empty launch (one byte read per block) unless blocks gave up at the gate.
template <class K1, class K>
int launch_parts(K1 kernel_one, K kernel, long grid, int threads, size_t lds, WaveParts& ps, int tiles_x) {
    Context& c = ctx();
    if constexpr (!std::is_same<K1, std::nullptr_t>::value) {
        if (ps.n == 1) {  // an ordinary call
            hipLaunchKernelGGL(kernel_one, dim3((unsigned)grid), dim3(threads), lds, c.compute, ps.a[0], tiles_x, ps.tiles_y[0], ps.run[0]);
            TOPO_HIP(hipGetLastError());
            return TOPO_AMD_OK;
        }
    }
    TOPO_REQUIRE(ps.gate.word == nullptr || (size_t)grid <= kGateSlots, "a gated launch of %ld blocks (at most %zu)", grid, kGateSlots);
    // (a kernel that serves ordinary calls too had its LDS size set by its launcher, once)
    if constexpr (!std::is_same<K1, std::nullptr_t>::value)
        TOPO_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(threads), lds, c.compute, ps, tiles_x);
    TOPO_HIP(hipGetLastError());
    if (ps.gate.word != nullptr && ps.gate.errors == nullptr) {  // (lean mode: no clean-up, see Context::gate_mode)
        TOPO_TRY(topo_amd_halo_wait());
        ps.gate.word = nullptr;
        ps.cleanup = 1;
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(threads), lds, c.compute, ps, tiles_x);
        TOPO_HIP(hipGetLastError());
    }
    return TOPO_AMD_OK;
}

template <int SIZE, int TH, int NWAVES, bool OUT_TPI, bool OUT_SUM, bool ALLOW_FRAC = false>
int launch_march(const Block& b, float* tpi_out, bool scaled = false) {
    using G = Geo<SIZE>;
    Context& c = ctx();
    WaveArgs a{b.in, tpi_out, nullptr, b.in_rows, b.in_row0, b.gny, b.nx, b.out_row0, b.out_rows,
               nullptr, nullptr, nullptr, 0, 0};
    constexpr size_t lds = (size_t)((TH + SIZE) + NWAVES) * ROWW * sizeof(int) + 16;
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        TOPO_HIP(hipFuncSetAttribute((const void*)tpi_march_kernel<SIZE, TH, NWAVES, OUT_TPI, OUT_SUM, ALLOW_FRAC>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int nblk = 0;
        TOPO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &nblk, (const void*)tpi_march_kernel<SIZE, TH, NWAVES, OUT_TPI, OUT_SUM, ALLOW_FRAC>, NWAVES * 64, lds));
        blocks_per_cu = nblk < 1 ? 1 : nblk;
    }
    WaveParts ps;
    int tiles_x = 0;
    long ntiles = 0;
    a.scaled = ALLOW_FRAC && scaled ? 1 : 0;
    a.report = ALLOW_FRAC && scaled ? dem_memo_report(b) : nullptr;
    TOPO_TRY(make_parts(b, a, TH, G::TILE_W, true, OUT_SUM && a.scaled == 0, &ps, &tiles_x, &ntiles));
    const long grid = march_grid(c, blocks_per_cu, ntiles);
    deal_parts(&ps, tiles_x, grid, blocks_per_cu);
    return launch_parts(tpi_march_kernel<SIZE, TH, NWAVES, OUT_TPI, OUT_SUM, ALLOW_FRAC>, tpi_march_kernel_parts<SIZE, TH, NWAVES, OUT_TPI, OUT_SUM, ALLOW_FRAC>, grid, NWAVES * 64, lds, ps, tiles_x);
}

// the first of the three marching passes of STD on fractional elevations: sums of trunc(x) for every tile, TPI of the whole-metre ones
template <int SIZE, int TH, int NWAVES, bool OUT_TPI>
int launch_march_sums(const Block& b, float* tpi_out) {
    using G = Geo<SIZE>;
    Context& c = ctx();
    WaveArgs a{b.in, tpi_out, nullptr, b.in_rows, b.in_row0, b.gny, b.nx, b.out_row0, b.out_rows,
               nullptr, nullptr, nullptr, 0, 0};
    constexpr size_t lds = (size_t)((TH + SIZE) + NWAVES) * ROWW * sizeof(int) + 16;
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        TOPO_HIP(hipFuncSetAttribute((const void*)tpi_march_sums_kernel<SIZE, TH, NWAVES, OUT_TPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int nblk = 0;
        TOPO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)tpi_march_sums_kernel<SIZE, TH, NWAVES, OUT_TPI>, NWAVES * 64, lds));
        blocks_per_cu = nblk < 1 ? 1 : nblk;
    }
    WaveParts ps;
    int tiles_x = 0;
    long ntiles = 0;
    TOPO_TRY(make_parts(b, a, TH, G::TILE_W, true, true, &ps, &tiles_x, &ntiles));
    const long grid = march_grid(c, blocks_per_cu, ntiles);
    deal_parts(&ps, tiles_x, grid, blocks_per_cu);
    return launch_parts(nullptr, tpi_march_sums_kernel<SIZE, TH, NWAVES, OUT_TPI>, grid, NWAVES * 64, lds, ps, tiles_x);
}

// ---- TPI on tiles with fractional elevations: the fraction pass ---------------------------------
// For the tiles tpi_march_kernel<.., ALLOW_FRAC> marked kNeedsFraction, p.sums holds the exact sum of
// trunc(x) over the disc.  This kernel marches the prefix sums of the fractional parts (integers in
// units of 2^-16 m) down the same runs and finalises
//   TPI = x - ((sum trunc(x) + 2^-16 sum g) - x) / (n - 1)
// in float64, which is the expression and the operands of the general kernel's fractional path
// (s1 = (su + c m) + sf with su + c m = sum trunc(x) exactly), so both give the same bits.  A
// window is carried only from a tile this kernel processed itself; otherwise it is staged in full.
// Each row's sum of trunc(x) arrives by an LDS-DMA load into the wave's idle slot of the segment
// totals, in flight during the chain; the pixel's own x is an ordinary load issued before the chain.
// WANT_STD (the third marching pass of STD on fractional elevations): p.sums2 / p.tile_c hold sum (trunc(x) - c)^2 and c of the
// tiles std_march_kernel<.., FRAC_STORE> went through; the pixel is finalised with the general kernel's choice and expressions
// (the integer form when the window's fractional sum is exactly 0, else s1 = sum trunc(x) + 2^-16 sum g and
// s2 = Su2 + 2 c Su + c^2 n in float64: exact integers whatever c is), hence with its bits.  TPI is written when p.tpi is set.
template <int SIZE, int TH, int NWAVES, bool WANT_STD = false>
__device__ __forceinline__ void tpi_fraction_march_kernel_body(const WaveArgs& p, int tiles_x, int tiles_y, const PartRun deal, const int vb0, const int nb) {
    using G = Geo<SIZE>;
    static_assert(G::T.centre == 0, "odd disc sizes only: the zeroed tap is the pixel itself");
    static_assert(TH % NWAVES == 0, "rows must split evenly over the waves");
    constexpr int NROWS = TH + SIZE - 1;
    constexpr int RW = TH / NWAVES;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    uint32_t* Q = lds_u;
    uint32_t* TOT = Q + (NROWS + 1) * ROWW;
    int* flag_word = reinterpret_cast<int*>(Q + (NROWS + 1 + NWAVES) * ROWW);
    uint32_t* TOT2 = reinterpret_cast<uint32_t*>(flag_word + 4);  // WANT_STD: a second landing row per wave (the sums of u^2)
    if (threadIdx.x == 0) *flag_word = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ntiles = tiles_x * tiles_y;
    const int vb = (vb0 + deal.shift) % nb;
    // the block's contiguous run of the strip-major tile list (deal_parts)
    const int first = deal.first(vb);
    const int last = min(first + deal.count(vb), ntiles);
    const double inv_nm1 = 1.0 / ((double)G::T.taps - 1.0);
    const double nd = (double)G::T.taps, inv_n = 1.0 / nd, inv_nn1 = 1.0 / (nd * (nd - 1.0));

    // Nothing to do on a DEM of whole metres: find that out with one flag per lane (64 tiles per
    // load) instead of walking the run with a dependent byte load per tile.
    bool any = false;
    for (int base = first; base < last; base += 64) {
        const int mine = base + lane;
        any = any || __builtin_amdgcn_ballot_w64(mine < last && p.defer[mine < last ? mine : first] == kNeedsFraction) != 0;
    }
    if (!any) return;  // the same for every thread of the block

    bool carry = false;
#pragma unroll 1
    for (int tile = first; tile < last; ++tile) {
        if (p.defer[tile] != kNeedsFraction) {
            carry = false;
            continue;
        }
        const int ty = tile % tiles_y;
        const int ox0 = (tile / tiles_y) * G::TILE_W;
        const int oy0 = (p.out_row0 / TH + ty) * TH;
        const int gx = ox0 - G::X0 + lane * NC;
        const int gy0 = oy0 + G::T.off_min;
        if (ty == 0) carry = false;
        if (carry) {
#ifdef MARCH_STAMPS
            long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#endif
            (void)stage_march<SIZE, TH, NWAVES, 1>(p, Q, flag_word, gy0, gx MARCH_STAMP_PASS);
        } else {
            __syncthreads();  // the previous tile's image is done with
            (void)stage_prefix<SIZE, TH, NWAVES, kStF, uint32_t>(p, lds_u, flag_word, gy0, gx, 0.0f, 0, 0.0f, 0.0f);
        }
        if (threadIdx.x == 0) p.defer[tile] = kTileDone;
        carry = true;
        const int ci = WANT_STD ? __builtin_amdgcn_readfirstlane(p.tile_c[tile]) : 0;

#pragma unroll 1
        for (int k = 0; k < RW; ++k) {
            const int jj = wave + k * NWAVES;
            const int oy = oy0 + jj;
            const int ocol = ox0 + lane * NC;
            const bool live = lane < G::NVL && ocol < p.nx && oy >= p.out_row0 && oy < p.out_row0 + p.out_rows;
            const size_t o = live ? (size_t)(oy - p.out_row0) * p.nx + ocol : 0;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.sums + o),
                                             (__attribute__((address_space(3))) void*)(TOT + wave * ROWW + lane * NC), 16,
                                             0, 0);
            if (WANT_STD)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.sums2 + o),
                                                 (__attribute__((address_space(3))) void*)(TOT2 + wave * ROWW + lane * NC), 16, 0, 0);
            const size_t xi = live ? (size_t)(oy - p.in_row0) * p.nx + ocol : 0;
            Vec4<float> xs{{0.f, 0.f, 0.f, 0.f}};
            if (!WANT_STD || p.tpi != nullptr) xs = *reinterpret_cast<const Vec4<float>*>(p.in + xi);
            uint32_t acc[NC];  // sum of the fractional parts in units of 2^-16 m (fits int32)
            wave_disc_sum<SIZE, uint32_t, 0, (SIZE >= 41)>(Q, jj, lane, acc);
            int lcol = lane * NC;
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(lcol) : : "memory");  // the DMA has landed
            const Vec4<int> sv = *reinterpret_cast<const Vec4<int>*>(TOT + wave * ROWW + lcol);
            Vec4<uint32_t> s2v{{0u, 0u, 0u, 0u}};
            if (WANT_STD) s2v = *reinterpret_cast<const Vec4<uint32_t>*>(TOT2 + wave * ROWW + lcol);
            if (!live) continue;
            Vec4<float> out_t, out_s;
#pragma unroll
            for (int t = 0; t < NC; ++t) {
                const double sf = (double)(int)acc[t] * (1.0 / 65536.0);
                const double s1 = (double)sv.v[t] + sf;
                const double x_ctr = (double)xs.v[t];
                out_t.v[t] = (float)((double)xs.v[t] - (s1 - x_ctr) * inv_nm1);
                if (WANT_STD) {
                    const int su = sv.v[t] - ci * G::T.taps;  // sum of u: what the general kernel's first chain yields
                    if ((int)acc[t] == 0) {
                        out_s.v[t] = std_from_int_sums(su, (uint64_t)s2v.v[t], (uint32_t)G::T.taps, (float)inv_nn1);
                    } else {
                        const double cd = (double)ci;
                        const double s2 = (double)s2v.v[t] + 2.0 * cd * (double)su + cd * cd * nd;
                        out_s.v[t] = std_from_sums(s1, s2, inv_n, inv_nm1);
                    }
                }
            }
            if (!WANT_STD || p.tpi != nullptr) *reinterpret_cast<Vec4<float>*>(p.tpi + o) = out_t;
            if (WANT_STD) *reinterpret_cast<Vec4<float>*>(p.sd + o) = out_s;
        }
    }
}

template <int SIZE, int TH, int NWAVES, bool WANT_STD = false>
__global__ __launch_bounds__(NWAVES * 64) void tpi_fraction_march_kernel(WaveParts ps, int tiles_x) {
    TOPO_RUN_PARTS_LOOP((tpi_fraction_march_kernel_body<SIZE, TH, NWAVES, WANT_STD>));
}

template <int SIZE, int TH, int NWAVES, bool WANT_STD = false>
int launch_fraction_march(const Block& b, float* tpi_out, float* std_out = nullptr) {
    using G = Geo<SIZE>;
    Context& c = ctx();
    WaveArgs a{b.in, tpi_out, std_out, b.in_rows, b.in_row0, b.gny, b.nx, b.out_row0, b.out_rows,
               nullptr, nullptr, nullptr, 0, 0};
    constexpr size_t lds = (size_t)((TH + SIZE) + NWAVES) * ROWW * sizeof(int) + 16 + (WANT_STD ? (size_t)NWAVES * ROWW * sizeof(int) : 0);
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        TOPO_HIP(hipFuncSetAttribute((const void*)tpi_fraction_march_kernel<SIZE, TH, NWAVES, WANT_STD>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int nblk = 0;
        TOPO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &nblk, (const void*)tpi_fraction_march_kernel<SIZE, TH, NWAVES, WANT_STD>, NWAVES * 64, lds));
        blocks_per_cu = nblk < 1 ? 1 : nblk;
    }
    WaveParts ps;
    int tiles_x = 0;
    long ntiles = 0;
    TOPO_TRY(make_parts(b, a, TH, G::TILE_W, true, true, &ps, &tiles_x, &ntiles));
    const long grid = march_grid(c, blocks_per_cu, ntiles);  // the same grid, hence the same runs, as launch_march
    deal_parts(&ps, tiles_x, grid, blocks_per_cu);
    return launch_parts(nullptr, tpi_fraction_march_kernel<SIZE, TH, NWAVES, WANT_STD>, grid, NWAVES * 64, lds, ps, tiles_x);
}

// ---- TPI on tiles with fractional elevations: the scaled one-chain route (round 4) -----------------------------------
// The exact route above costs a fractional DEM two stagings and two chains per tile (9.6 ms at 67 px on 32768^2 against
// 4.4 ms on whole metres).  The reference squares trunc(x) (topo.py:300) but TPI only needs s1 = sum x (topo.py:175-181),
// and ONE integer chain carries it when x is taken in units of 2^-8 m:  q = rint(256 x), absolute - no offset, so a
// sample's q does not depend on the tile, run or row block that stages it, sums of q are exact integers, and row blocks
// keep their bit-identity.  The prefix image and the chain work modulo 2^32 like the whole-metre kernel's; the true
// sum S over the disc's n taps lies within n 256 (relief of the window) of n q_centre, so
//     S = n q_ctr + int32(S mod 2^32 - n q_ctr mod 2^32)
// is exact whenever n (max q - min q over the staged window) < 2^31: 2457 m of relief at 67 px.  The block forms that
// range per tile (block_range; for a marched window from the ranges of the tiles whose rows it holds) and leaves a tile
// that fails the test - nodata next to terrain - to the general kernel, which is exact, like every tile whose window
// reaches over the DEM's edge (zeros are no terrain).  Only such a failing window can make a row block differ from the
// whole DEM (the block sees less of it); everywhere else row blocks give the single block's bits.  TPI = x - (S / 256 - x) / (n - 1) with the pixel's own x read from the DEM.
// Error: |rint(256 x) / 256 - x| <= 2^-9 m = 1.95 mm per sample, hence <= 1.95 mm on the mean and on TPI (about 0.03 mm rms
// at 67 px when the fractional parts are spread evenly; the bound is reached by a DEM whose samples all carry one
// fractional part off the 2^-8 m grid), against the 1.4 - 1.7 mm the reference's own float32 FFT is off by
// and the 1e-4 x range of the contract.  A pixel whose disc holds whole metres only gets the whole-metre kernel's bits.
// TOPO_AMD_TPI_FRACTION_EXACT=1: the exact two-pass route above instead.
// TAKE_ALL: every tile, classified here (non-finite or absurd samples: the general kernel) - the first and only marching
// launch on a DEM the library remembers as mostly fractional (dem_memo): the whole-metre launch in front of it would
// stage every tile only to find it fractional (7.1 ms against 5 ms at 67 px on 32768^2).  A window of whole metres gets the
// whole-metre kernel's bits here too (S = 256 sum trunc(x) exactly).
// The unit of the scaled route: the largest power of two between 2^8 and 2^16 that keeps n x (value range of the raster) x
// unit below 2^30 and |x| unit below 2^28 - from the raster class (common.hpp: a lattice sample of the WHOLE raster, the same
// for every row block of it).  An ordinary DEM in metres gets 2^8 (error <= 2^-9 m per sample); a raster of small values -
// kilometres, a normalised surface - gets finer units, so that the error stays below 4e-6 of its value range whatever the
// range is.  (The tile-by-tile test in the kernel guards the exactness of the unwrapping and - scaled_abs_lim - the float to
// integer conversion of every staged sample; this only sets the precision.)
inline float scaled_unit(int taps) {
    const RasterClass c = current_class();
    int k = 8;
    if (c.lo <= c.hi) {
        const double range = std::max((double)c.hi - (double)c.lo, 1e-30);
        const double mag = std::max(std::max(std::fabs((double)c.lo), std::fabs((double)c.hi)), 1e-30);
        const int k_range = (int)std::floor(std::log2(1073741824.0 / ((double)taps * range)));
        const int k_mag = (int)std::floor(std::log2(268435456.0 / mag));
        k = std::max(8, std::min(16, std::min(k_range, k_mag)));
    }
    return std::ldexp(1.0f, k);
}

template <int SIZE, int TH, int NWAVES, bool TAKE_ALL = false>
__device__ __forceinline__ void tpi_scaled_march_kernel_body(const WaveArgs& p, int tiles_x, int tiles_y, const PartRun deal, const int vb0, const int nb) {
    using G = Geo<SIZE>;
    static_assert(G::T.centre == 0, "odd disc sizes only: the zeroed tap is the pixel itself");
    static_assert(TH % NWAVES == 0, "rows must split evenly over the waves");
    constexpr int NROWS = TH + SIZE - 1;
    constexpr int RW = TH / NWAVES;
    constexpr int kHist = (SIZE - 1 + TH - 1) / TH;  // tiles above whose rows a marched window still holds
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    uint32_t* Q = lds_u;
    int* flag_word = reinterpret_cast<int*>(Q + (NROWS + 1 + NWAVES) * ROWW);
    if (threadIdx.x == 0) {
        flag_word[0] = 0;
        flag_word[2] = 0x7fffffff;
        flag_word[3] = -0x7fffffff - 1;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ntiles = tiles_x * tiles_y;
    const int vb = (vb0 + deal.shift) % nb;
    const int first = deal.first(vb);
    const int last = min(first + deal.count(vb), ntiles);
    const double inv_nm1 = 1.0 / ((double)G::T.taps - 1.0);

    if (!TAKE_ALL) {
        // nothing to do on a DEM of whole metres: one flag per lane, 64 tiles per load
        bool any = false;
        for (int base = first; base < last; base += 64) {
            const int mine = base + lane;
            any = any || __builtin_amdgcn_ballot_w64(mine < last && p.defer[mine < last ? mine : first] == kNeedsFraction) != 0;
        }
        if (!any) return;  // the same for every thread of the block
    }

    const double inv_unit = 1.0 / (double)p.unit;  // a power of two
    int* rowlo = flag_word + 4;  // the row-by-row test (below): range of every staged row, 2 x 192 words, then the answer
    bool carry = false;
    int hlo[kHist + 1], hhi[kHist + 1];  // range of the new rows of this tile ([0]) and of the tiles above it
#pragma unroll
    for (int k = 0; k <= kHist; ++k) hlo[k] = 0x7fffffff, hhi[k] = -0x7fffffff - 1;
    int seen_frac = 0;  // TAKE_ALL: tiles of the run whose new rows hold a fractional sample (the report)
#pragma unroll 1
    for (int tile = first; tile < last; ++tile) {
        if (!TAKE_ALL && p.defer[tile] != kNeedsFraction) {
            carry = false;
            continue;
        }
        const int ty = tile % tiles_y;
        const int ox0 = (tile / tiles_y) * G::TILE_W;
        const int oy0 = (p.out_row0 / TH + ty) * TH;
        const int gx = ox0 - G::X0 + lane * NC;
        const int gy0 = oy0 + G::T.off_min;
        if (ty == 0) carry = false;
        // A tile whose window reaches over the DEM's edge goes to the general kernel unstaged: its padded taps are
        // zeros, n q_ctr is no estimate of such a sum, and the test on the range would depend on how much of the window a
        // row block has in view.  (Geometry of the whole DEM only: every row block takes the same decision.  The two outer
        // strips and the first and last tile row: 1.5 % of the tiles of a 32768^2 DEM.)
        if (gy0 < 0 || gy0 + NROWS > p.gny || ox0 - G::X0 < 0 || ox0 - G::X0 + ROWW > p.nx) {
            if (threadIdx.x == 0) p.defer[tile] = kTileGeneral;
            carry = false;
            continue;
        }
        int range[2];
        int flags = 0;
        if (carry) {
#ifdef MARCH_STAMPS
            long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#endif
            flags = stage_march<SIZE, TH, NWAVES, 3>(p, Q, flag_word, gy0, gx MARCH_STAMP_PASS, range);
#pragma unroll
            for (int k = kHist; k >= 1; --k) hlo[k] = hlo[k - 1], hhi[k] = hhi[k - 1];
            hlo[0] = __builtin_amdgcn_readfirstlane(range[0]);
            hhi[0] = __builtin_amdgcn_readfirstlane(range[1]);
        } else {
            __syncthreads();  // the previous tile's image is done with
            flags = stage_prefix<SIZE, TH, NWAVES, kStS, uint32_t, true>(p, lds_u, flag_word, gy0, gx, p.unit, 0, 0.0f, 0.0f, range);
#pragma unroll
            for (int k = 0; k <= kHist; ++k) {  // the whole window was staged: its range stands for every tile in it
                hlo[k] = __builtin_amdgcn_readfirstlane(range[0]);
                hhi[k] = __builtin_amdgcn_readfirstlane(range[1]);
            }
        }
        int wlo = hlo[0], whi = hhi[0];
#pragma unroll
        for (int k = 1; k <= kHist; ++k) wlo = min(wlo, hlo[k]), whi = max(whi, hhi[k]);
        // the unwrapping below is exact for every pixel of the tile when n x (range of the window) < 2^31
        const bool fits = (long long)G::T.taps * ((long long)whi - (long long)wlo) < (1ll << 31);
        // a non-finite or absurd sample in the window: like a window with too much relief, it is taken row by row - the rows
        // whose own windows are free of such samples are this kernel's (the sums are integers modulo 2^32: what a bad sample
        // leaves in the prefix image cancels in every difference that does not span it), the others the general kernel's
        const bool leave = (flags & kTileFloat) != 0;
        if (TAKE_ALL && (flags & kTileFrac)) ++seen_frac;
        // The window as a whole holds too much relief (nodata next to terrain): the test is made row by row, each output
        // row on ITS OWN window - rows oy - M ... oy + M of the strip's staged columns, which every row block that computes
        // the row holds - so that which rows go to the exact general kernel is a function of the data and the global
        // grid, not of the rows a block happens to have in view (when the whole window passes, every row's does).
        uint64_t rows_out = 0;
        if (!fits || leave) {
            for (int r = wave; r < NROWS; r += NWAVES) {
                const bool ok = row4_inside(p, gy0 + r, gx);
                const Vec4<float> v = load_row4(p, gy0 + r, gx);
                int lo = 0x7fffffff, hi = -0x7fffffff - 1;
#pragma unroll
                for (int s2 = 0; s2 < NC; ++s2) {
                    const int q = (int)stage_value<kStS>(v.v[s2], p.unit, 0);
                    // (a sample the integer chain cannot take - non-finite, beyond 2^18 or what the unit leaves of 2^30 - counts
                    // as the widest range)
                    const bool bad = !(fabsf(truncf(v.v[s2])) <= scaled_abs_lim(p.unit));
                    lo = ok ? (bad ? -0x7fffffff - 1 : min(lo, q)) : lo;
                    hi = ok ? (bad ? 0x7fffffff : max(hi, q)) : hi;
                }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    lo = min(lo, __shfl_xor(lo, m));
                    hi = max(hi, __shfl_xor(hi, m));
                }
                if (lane == 0) {
                    rowlo[r] = lo;
                    rowlo[192 + r] = hi;
                }
            }
            __syncthreads();
            static_assert(TH <= 64 && NROWS <= 192, "one ballot holds the tile's rows");
            if (wave == 0) {
                bool fail = false;
                if (lane < TH) {
                    int lo = 0x7fffffff, hi = -0x7fffffff - 1;
                    for (int k = 0; k < SIZE; ++k) {
                        lo = min(lo, rowlo[lane + k]);
                        hi = max(hi, rowlo[192 + lane + k]);
                    }
                    fail = lo <= hi && (long long)G::T.taps * ((long long)hi - (long long)lo) >= (1ll << 31);
                }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(fail);
                if (lane == 0) {
                    rowlo[384] = (int)(uint32_t)m;
                    rowlo[385] = (int)(uint32_t)(m >> 32);
                }
            }
            __syncthreads();
            rows_out = ((uint64_t)(uint32_t)rowlo[385] << 32) | (uint32_t)rowlo[384];
            rows_out = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(rows_out >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)rows_out);
        }
        if (threadIdx.x == 0) {
            p.defer[tile] = rows_out != 0 ? kTileGeneralRows : kTileDone;
            if (rows_out != 0) p.rowmask[tile] = rows_out;
            flag_word[0] = 0;
            flag_word[2] = 0x7fffffff;  // (every thread has read flags and range; the next atomics are behind a barrier)
            flag_word[3] = -0x7fffffff - 1;
        }
        carry = !leave;  // (a window with such a sample is not carried on: the next tile is staged and classified in full)

#pragma unroll 1
        for (int k = 0; k < RW; ++k) {
            const int jj = wave + k * NWAVES;
            const int oy = oy0 + jj;
            if (oy < p.out_row0 || oy >= p.out_row0 + p.out_rows || ((rows_out >> jj) & 1)) continue;
            const int ocol = ox0 + lane * NC;
            const bool live = lane < G::NVL && ocol < p.nx;
            const size_t xi = live ? (size_t)(oy - p.in_row0) * p.nx + ocol : 0;
            const Vec4<float> xs = *reinterpret_cast<const Vec4<float>*>(p.in + xi);
            uint32_t acc[NC];  // sum of q over the disc modulo 2^32
            wave_disc_sum<SIZE, uint32_t, 0, (SIZE >= 41)>(Q, jj, lane, acc);
            if (!live) continue;
            // the pixel's own q: prefix through its row minus prefix above it
            const uint32_t* own = Q + (jj - G::T.off_min) * ROWW + lane * NC + G::X0;
            const Vec4<uint32_t> hi = *reinterpret_cast<const Vec4<uint32_t>*>(own + ROWW);
            const Vec4<uint32_t> lo = *reinterpret_cast<const Vec4<uint32_t>*>(own);
            Vec4<float> out_t;
#pragma unroll
            for (int t = 0; t < NC; ++t) {
                const uint32_t qc = hi.v[t] - lo.v[t];
                const int off = (int)(acc[t] - (uint32_t)G::T.taps * qc);  // S - n q_ctr, exact (see above)
                const double s1 = ((double)G::T.taps * (double)(int)qc + (double)off) * inv_unit;
                const double x = (double)xs.v[t];
                out_t.v[t] = (float)(x - (s1 - x) * inv_nm1);
            }
            *reinterpret_cast<Vec4<float>*>(p.tpi + (size_t)(oy - p.out_row0) * p.nx + ocol) = out_t;
        }
    }
    if (TAKE_ALL && p.report != nullptr && vb == nb / 2 && threadIdx.x == 0) {
        __hip_atomic_store(p.report + 1, (uint32_t)seen_frac, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(p.report, (uint32_t)(last > first ? last - first : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int SIZE, int TH, int NWAVES, bool TAKE_ALL = false>
__global__ __launch_bounds__(NWAVES * 64) void tpi_scaled_march_kernel(WaveParts ps, int tiles_x) {
    TOPO_RUN_PARTS_LOOP((tpi_scaled_march_kernel_body<SIZE, TH, NWAVES, TAKE_ALL>));
}

template <int SIZE, int TH, int NWAVES, bool TAKE_ALL = false>
int launch_scaled_march(const Block& b, float* tpi_out) {
    using G = Geo<SIZE>;
    Context& c = ctx();
    WaveArgs a{b.in, tpi_out, nullptr, b.in_rows, b.in_row0, b.gny, b.nx, b.out_row0, b.out_rows,
               nullptr, nullptr, nullptr, 0, 0};
    a.scaled = 1;
    a.unit = scaled_unit(G::T.taps);
    a.report = TAKE_ALL ? dem_memo_report(b) : nullptr;
    constexpr size_t lds = (size_t)((TH + SIZE) + NWAVES) * ROWW * sizeof(int) + 16 + 388 * sizeof(int);
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        TOPO_HIP(hipFuncSetAttribute((const void*)tpi_scaled_march_kernel<SIZE, TH, NWAVES, TAKE_ALL>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int nblk = 0;
        TOPO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &nblk, (const void*)tpi_scaled_march_kernel<SIZE, TH, NWAVES, TAKE_ALL>, NWAVES * 64, lds));
        blocks_per_cu = nblk < 1 ? 1 : nblk;
    }
    WaveParts ps;
    int tiles_x = 0;
    long ntiles = 0;
    TOPO_TRY(make_parts(b, a, TH, G::TILE_W, true, false, &ps, &tiles_x, &ntiles));
    const long grid = march_grid(c, blocks_per_cu, ntiles);  // the same grid, hence the same runs, as launch_march
    deal_parts(&ps, tiles_x, grid, blocks_per_cu);
    return launch_parts(nullptr, tpi_scaled_march_kernel<SIZE, TH, NWAVES, TAKE_ALL>, grid, NWAVES * 64, lds, ps, tiles_x);
}

// ---- STD on tiles of whole metres: the second marching kernel ------------------------------------
// STD needs s1 = sum trunc(x) and s2 = sum trunc(x)^2 over the disc.  tpi_march_kernel<.., OUT_SUM>
// leaves s1 in p.sums (exact, no offset).  This kernel marches the prefix sums of u^2,
// u = trunc(x) - c, down the same runs of the same tile list and finalises:
//   s2 = Su2 + 2 c (s1 - c m) + c^2 m        (m in-domain taps; every term an exact integer < 2^53)
// so the float64 variance is the one the general kernel computes, bit for bit, whatever c either
// of them used.  c is constant along a carried run (a changed c would change every carried u^2);
// the 32-bit sums need |u| <= lim32 over the whole window (3409 u^2 < 2^32).  When the rows a tile
// adds break that, the window is restaged in full around the tile's own centre value; a tile that
// is wide even then is marked for the general kernel, which runs two 16-bit half chains on it.
// Tiles the first kernel marked are skipped.
template <int SIZE, int TH, int NWAVES, bool FULL>
__device__ __forceinline__ int stage_u2(const WaveArgs& p, uint32_t* Q, int* flag_word, int gy0, int gx, float c,
                                        int ci, float lim32) {
    constexpr int NROWS = TH + SIZE - 1;
    constexpr int KEEP = NROWS + 1 - TH;
    constexpr int NT = NWAVES * 64;
    constexpr int SL = FULL ? (NROWS + NWAVES - 1) / NWAVES : TH / NWAVES;  // rows staged per wave
    constexpr int FIRST = FULL ? 0 : KEEP - 1;                              // first window row staged
    constexpr int MOVE4 = KEEP * 64;
    constexpr int MOVES = (MOVE4 + NT - 1) / NT;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    uint32_t* TOT = Q + (NROWS + 1) * ROWW;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r0 = FIRST + wave * SL;
    Vec4<float> v[SL];
#pragma unroll
    for (int k = 0; k < SL; ++k) v[k] = load_row4(p, gy0 + r0 + k, gx);
    if (!FULL) {
        u4 m[MOVES];
        const u4* src = reinterpret_cast<const u4*>(Q + TH * ROWW);
#pragma unroll
        for (int s = 0; s < MOVES; ++s) {
            const int idx = (int)threadIdx.x + s * NT;
            m[s] = src[idx < MOVE4 ? idx : MOVE4 - 1];
        }
        __syncthreads();  // every wave is done reading the previous tile's image
        u4* dst = reinterpret_cast<u4*>(Q);
#pragma unroll
        for (int s = 0; s < MOVES; ++s) {
            const int idx = (int)threadIdx.x + s * NT;
            if (idx < MOVE4) dst[idx] = m[s];
        }
    } else if (wave == 0) {
        *reinterpret_cast<Vec4<uint32_t>*>(Q + lane * NC) = Vec4<uint32_t>{{0u, 0u, 0u, 0u}};
    }
    uint32_t amax = 0, umax = 0;
    bool frac = false;
    Vec4<uint32_t> run{{0u, 0u, 0u, 0u}};
#pragma unroll
    for (int k = 0; k < SL; ++k) {
        const int r = r0 + k;
        if (FULL && r >= NROWS) break;
        const bool ok = row4_inside(p, gy0 + r, gx);
#pragma unroll
        for (int s = 0; s < NC; ++s) {
            const float x = v[k].v[s];
            const float t = truncf(x);
            const float d = t - c;  // exact: |t|, |c| <= kAbsLim on the tiles that stay here
            frac |= ok && (x != t);
            amax = max(amax, ok ? (__float_as_uint(t) & 0x7fffffffu) : 0u);
            umax = max(umax, ok ? (__float_as_uint(d) & 0x7fffffffu) : 0u);
            const uint32_t u = (uint32_t)((int)t - ci);
            run.v[s] += ok ? u * u : 0u;
        }
        *reinterpret_cast<Vec4<uint32_t>*>(Q + (r + 1) * ROWW + lane * NC) = run;
    }
    *reinterpret_cast<Vec4<uint32_t>*>(TOT + wave * ROWW + lane * NC) = run;
    int wf = 0;
    if (__builtin_amdgcn_ballot_w64(frac)) wf |= kTileFrac;
    if (__builtin_amdgcn_ballot_w64(amax > __float_as_uint(kAbsLim))) wf |= kTileFloat;
    if (__builtin_amdgcn_ballot_w64(umax > __float_as_uint(lim32))) wf |= kTileWide;
    if (lane == 0 && wf) atomicOr(flag_word, wf);
    __syncthreads();
    const int all = *flag_word;
    Vec4<uint32_t> off{{0u, 0u, 0u, 0u}};
    if (!FULL) off = *reinterpret_cast<const Vec4<uint32_t>*>(Q + (KEEP - 1) * ROWW + lane * NC);
    for (int w = 0; w < wave; ++w) {
        const Vec4<uint32_t> t = *reinterpret_cast<const Vec4<uint32_t>*>(TOT + w * ROWW + lane * NC);
#pragma unroll
        for (int s = 0; s < NC; ++s) off.v[s] += t.v[s];
    }
    if (!FULL || wave > 0) {
#pragma unroll
        for (int k = 0; k < SL; ++k) {
            const int r = r0 + k;
            if (FULL && r >= NROWS) break;
            Vec4<uint32_t>* q = reinterpret_cast<Vec4<uint32_t>*>(Q + (r + 1) * ROWW + lane * NC);
            Vec4<uint32_t> x = *q;
#pragma unroll
            for (int s = 0; s < NC; ++s) x.v[s] += off.v[s];
            *q = x;
        }
    }
    __syncthreads();
    return all;
}

// FRAC_STORE (the second of the three marching passes of STD on fractional elevations): a tile the first pass marked
// kNeedsFraction is gone through as well - the sums of u^2 do not care about the fractional parts - but not finalised: its sums
// go to p.sums2 and its offset to p.tile_c, for the fraction pass (tpi_fraction_march_kernel<.., WANT_STD>).  Such a tile at the
// DEM's border (the in-domain tap counts enter there) is left to the general kernel.
template <int SIZE, int TH, int NWAVES, bool FRAC_STORE = false>
__device__ __forceinline__ void std_march_kernel_body(const WaveArgs& p, int tiles_x, int tiles_y, const PartRun deal, const int vb0, const int nb) {
    using G = Geo<SIZE>;
    static_assert(TH % NWAVES == 0, "rows must split evenly over the waves");
    constexpr int NROWS = TH + SIZE - 1;
    constexpr int RW = TH / NWAVES;
    constexpr int kGiveUp = 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    uint32_t* Q = lds_u;
    uint32_t* TOT = Q + (NROWS + 1) * ROWW;  // segment totals while staging; one sums row per wave afterwards
    unsigned short* PL = reinterpret_cast<unsigned short*>(lds_u + (NROWS + 1 + NWAVES) * ROWW);
    int* flag_word = reinterpret_cast<int*>(PL + ((TH * (SIZE + 1) + 7) & ~7));
    if (threadIdx.x == 0) *flag_word = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ntiles = tiles_x * tiles_y;
    const int vb = (vb0 + deal.shift) % nb;
    // the block's contiguous run of the strip-major tile list (deal_parts)
    const int first = deal.first(vb);
    const int last = min(first + deal.count(vb), ntiles);
    const double n = (double)G::T.taps;
    const double inv_nn1 = 1.0 / (n * (n - 1.0));
    const float lim32 = floorf(sqrtf(4294967295.0f / (float)G::T.taps));

    bool carry = false;
    float c = 0.0f;
    int ci = 0;
    int deferred_in_a_row = 0;
#pragma unroll 1
    for (int tile = first; tile < last; ++tile) {
        const int state = __builtin_amdgcn_readfirstlane((int)p.defer[tile]);
        const bool fractional = FRAC_STORE && state == kNeedsFraction;
        if (state != 0 && !fractional) {  // left by the first kernel (fractional / non-finite / absurd samples)
            carry = false;
            continue;
        }
        if (deferred_in_a_row >= kGiveUp) {
            if (threadIdx.x == 0) p.defer[tile] = 1;
            continue;
        }
        const int ty = tile % tiles_y;
        const int ox0 = (tile / tiles_y) * G::TILE_W;
        const int oy0 = (p.out_row0 / TH + ty) * TH;
        const int gx = ox0 - G::X0 + lane * NC;
        const int gy0 = oy0 + G::T.off_min;
        const bool border = gy0 < 0 || gy0 + NROWS > p.gny || ox0 - G::X0 < 0 || ox0 - G::X0 + ROWW > p.nx;
        if (ty == 0) carry = false;
        if (fractional && border) {
            if (threadIdx.x == 0) p.defer[tile] = 1;
            carry = false;
            continue;
        }
        constexpr int kFlagMask = FRAC_STORE ? ~(int)kTileFrac : ~0;  // (fractional samples are what this pass is there for)
        int flags = kTileWide;
        if (carry) {
            flags = stage_u2<SIZE, TH, NWAVES, false>(p, Q, flag_word, gy0, gx, c, ci, lim32) & kFlagMask;
            if (threadIdx.x == 0) *flag_word = 0;  // read by every thread; the next atomicOr is behind a barrier
        }
        if (flags & kTileWide) {
            // (re)start: the whole window, around this tile's own centre value
            int cy = min(max(oy0 + TH / 2, 0), p.gny - 1);
            cy = min(max(cy, p.in_row0), p.in_row0 + p.in_rows - 1);
            const int cx = min(ox0 + G::TILE_W / 2, p.nx - 1);
            c = truncf(p.in[(size_t)(cy - p.in_row0) * p.nx + cx]);
            if (!(fabsf(c) <= kAbsLim)) c = 0.0f;
            // the same for every lane: keep it in a scalar register, the chain needs the vector ones
            c = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(c)));
            ci = (int)c;
            __syncthreads();  // the image and the flag word are free
            flags = stage_u2<SIZE, TH, NWAVES, true>(p, Q, flag_word, gy0, gx, c, ci, lim32) & kFlagMask;
            if (threadIdx.x == 0) *flag_word = 0;
        }
        if (flags != 0) {  // wide even around its own centre (or a class the first kernel would have caught)
            if (threadIdx.x == 0) p.defer[tile] = 1;
            ++deferred_in_a_row;
            carry = false;
            continue;
        }
        deferred_in_a_row = 0;
        carry = true;
        if (fractional) {
            // the sums of u^2 and the offset they were taken with: the fraction pass finalises the tile
            if (threadIdx.x == 0) p.tile_c[tile] = ci;
#pragma unroll 1
            for (int k = 0; k < RW; ++k) {
                const int jj = wave + k * NWAVES;
                const int oy = oy0 + jj;
                if (oy < p.out_row0 || oy >= p.out_row0 + p.out_rows) continue;
                uint32_t acc[NC];
                wave_disc_sum<SIZE, uint32_t, 0, true>(Q, jj, lane, acc);
                const int ocol = ox0 + lane * NC;
                if (lane < G::NVL && ocol < p.nx)
                    *reinterpret_cast<Vec4<uint32_t>*>(p.sums2 + (size_t)(oy - p.out_row0) * p.nx + ocol) = Vec4<uint32_t>{{acc[0], acc[1], acc[2], acc[3]}};
            }
            continue;
        }

        if (border) {
            if (threadIdx.x < TH) {
                // in-domain rows of each column run, prefix-summed over the column offsets
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
            __syncthreads();
        }
        // one copy of the row loop per kind of tile: away from the DEM border every pixel has all n
        // taps, so m is a constant and only the short formula is compiled in
        auto rows = [&](auto border_tag) {
            constexpr bool BORDER = decltype(border_tag)::value;
#pragma unroll 1
            for (int k = 0; k < RW; ++k) {
                const int jj = wave + k * NWAVES;
                const int oy = oy0 + jj;
                // The row's sums of x come from a plane that left L2 long ago.  Fetch them straight
                // into this wave's (now idle) slot of the segment totals with an LDS-DMA load: it
                // is in flight during the chain and costs no vector register, which this kernel
                // does not have to spare (a register prefetch spills; an ordinary load after the
                // chain exposes the whole memory latency once per row).
                {
                    const int pcol = ox0 + lane * NC;
                    const bool pl_ok = lane < G::NVL && pcol < p.nx && oy >= p.out_row0 && oy < p.out_row0 + p.out_rows;
                    const size_t po = pl_ok ? (size_t)(oy - p.out_row0) * p.nx + pcol : 0;
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(p.sums + po),
                        (__attribute__((address_space(3))) void*)(TOT + wave * ROWW + lane * NC), 16, 0, 0);
                }
                uint32_t acc[NC];
                wave_disc_sum<SIZE, uint32_t, 0, true>(Q, jj, lane, acc);
                // Everything below is derived from the lane's column again, behind an opaque barrier:
                // otherwise the per-lane invariants of the finalisation (column limits, output index)
                // are hoisted above the chain, and at 168 VGPRs the chain then loses the registers it
                // needs to keep its next pair of LDS reads in flight.
                int lcol = lane * NC;
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(lcol) : : "memory");  // the DMA has landed
                const Vec4<int> sv = *reinterpret_cast<const Vec4<int>*>(TOT + wave * ROWW + lcol);
                const int ocol = ox0 + lcol;
                const bool lane_ok = lcol < G::NVL * NC && ocol < p.nx;
                if (!lane_ok || oy < p.out_row0 || oy >= p.out_row0 + p.out_rows) continue;
                const size_t o = (size_t)(oy - p.out_row0) * p.nx + ocol;
                Vec4<float> out_s;
#pragma unroll
                for (int t = 0; t < NC; ++t) {
                    double m = n;
                    if (BORDER) {
                        const int d_lo = max(G::T.off_min, -(ocol + t));
                        const int d_hi = min(G::T.off_max, p.nx - 1 - (ocol + t));
                        const unsigned short* pl = PL + jj * (SIZE + 1) - G::T.off_min;
                        m = d_hi >= d_lo ? (double)((int)pl[d_hi + 1] - (int)pl[d_lo]) : 0.0;
                    }
                    if (!BORDER || m == n) {
                        // sum of u, what the general kernel's first chain yields: |ci n| < 2^30, |su| < 2^25
                        out_s.v[t] = std_from_int_sums(sv.v[t] - ci * G::T.taps, (uint64_t)acc[t], (uint32_t)G::T.taps, (float)inv_nn1);
                    } else {
                        out_s.v[t] = std_from_border_sums((int64_t)sv.v[t] - (int64_t)ci * (int)m, (uint64_t)acc[t], ci, (int)m,
                                                          (uint32_t)G::T.taps, (float)inv_nn1);
                    }
                }
                *reinterpret_cast<Vec4<float>*>(p.sd + o) = out_s;
            }
        };
        if (border) {
            rows(std::true_type{});
        } else {
            rows(std::false_type{});
        }
    }
}

template <int SIZE, int TH, int NWAVES, bool FRAC_STORE = false>
__global__ __launch_bounds__(NWAVES * 64) void std_march_kernel(WaveParts ps, int tiles_x) {
    TOPO_RUN_PARTS_LOOP((std_march_kernel_body<SIZE, TH, NWAVES, FRAC_STORE>));
}

template <int SIZE, int TH, int NWAVES, bool FRAC_STORE = false>
int launch_std_march(const Block& b, float* std_out) {
    using G = Geo<SIZE>;
    Context& c = ctx();
    WaveArgs a{b.in, nullptr, std_out, b.in_rows, b.in_row0, b.gny, b.nx, b.out_row0, b.out_rows,
               nullptr, nullptr, nullptr, 0, 0};
    constexpr size_t lds = (size_t)((TH + SIZE) + NWAVES) * ROWW * sizeof(int) +
                           (size_t)((TH * (SIZE + 1) + 7) & ~7) * sizeof(unsigned short) + 16;
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        TOPO_HIP(hipFuncSetAttribute((const void*)std_march_kernel<SIZE, TH, NWAVES, FRAC_STORE>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int nblk = 0;
        TOPO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)std_march_kernel<SIZE, TH, NWAVES, FRAC_STORE>,
                                                              NWAVES * 64, lds));
        blocks_per_cu = nblk < 1 ? 1 : nblk;
    }
    WaveParts ps;
    int tiles_x = 0;
    long ntiles = 0;
    TOPO_TRY(make_parts(b, a, TH, G::TILE_W, true, true, &ps, &tiles_x, &ntiles));
    const long grid = march_grid(c, blocks_per_cu, ntiles);  // the same grid, hence the same runs, as launch_march
    deal_parts(&ps, tiles_x, grid, blocks_per_cu);
    return launch_parts(nullptr, std_march_kernel<SIZE, TH, NWAVES, FRAC_STORE>, grid, NWAVES * 64, lds, ps, tiles_x);
}

// only_deferred: process the tiles a preceding launch_march of the same geometry marked.
// map_th: tile height of the marching launch when it differs from TH.
template <int SIZE, int TH, int NWAVES, bool WANT_TPI, bool WANT_STD>
int launch_wave(const Block& b, float* tpi_out, float* std_out, bool only_deferred = false, int map_th = 0,
                int map_tw = 0) {
    using G = Geo<SIZE>;
    Context& c = ctx();
    WaveArgs a{b.in, tpi_out, std_out, b.in_rows, b.in_row0, b.gny, b.nx, b.out_row0, b.out_rows,
               nullptr, nullptr, nullptr, 0, 0};
    constexpr size_t lds = (size_t)((TH + SIZE) + NWAVES) * ROWW * sizeof(float) +
                           (size_t)((TH * (SIZE + 1) + 7) & ~7) * sizeof(unsigned short) + 16;
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    static_assert(SIZE * SIZE < 65536, "tap counts must fit the 16-bit border table");
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        TOPO_HIP(hipFuncSetAttribute((const void*)disc_wave_kernel<SIZE, TH, NWAVES, WANT_TPI, WANT_STD>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int nblk = 0;
        TOPO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &nblk, (const void*)disc_wave_kernel<SIZE, TH, NWAVES, WANT_TPI, WANT_STD>, NWAVES * 64, lds));
        blocks_per_cu = nblk < 1 ? 1 : nblk;
    }
    WaveParts ps;
    int tiles_x = 0;
    long ntiles = 0;
    TOPO_TRY(make_parts(b, a, TH, G::TILE_W, only_deferred, false, &ps, &tiles_x, &ntiles, only_deferred ? map_th : 0,
                        only_deferred ? map_tw : 0));
    // persistent blocks fill the chip (whole XCD rounds: the tile list is cut into XCD-contiguous runs)
    const long grid = march_grid(c, blocks_per_cu, ntiles);
    void* scratch = nullptr;
    TOPO_TRY(workspace(2, (size_t)grid * kScratchPlanes * TH * ROWW * sizeof(uint32_t), &scratch));
    for (int k = 0; k < kMaxParts; ++k) ps.a[k].scratch = (uint32_t*)scratch;
    deal_parts(&ps, tiles_x, grid, blocks_per_cu);  // (this kernel deals its tiles round-robin: only the shifts matter)
    return launch_parts(nullptr, disc_wave_kernel<SIZE, TH, NWAVES, WANT_TPI, WANT_STD>, grid, NWAVES * 64, lds, ps, tiles_x);
}

}  // namespace

}  // namespace topo

#include "disc_ring_impl.hpp"

namespace topo {

namespace {

// Tile heights: as tall as LDS allows (prefix rows + segment totals + border table <= 160 KiB),
// capped at the heights that measured best for 67 px.
constexpr int tile_rows(int size, int nwaves, int cap) {
    int th = cap;
    while (th > nwaves && (size_t)(th + size + nwaves) * ROWW * 4 + (size_t)(th * (size + 1) + 8) * 2 + 16 > 160 * 1024)
        th -= nwaves;
    return th;
}

// Which discs take the marching kernels.  Measured on a 32768^2 DEM of whole metres
// (profiles/r01_std_crossover.txt): for STD / TPI+STD the marching pair wins
// from 31 px (2 %) to 67 px (11 %) and loses below (three planes of traffic in one general kernel
// against five) and wherever LDS no longer holds the full 60-row tile (101 px: 36-row tiles, 46 ms
// against 31 ms).  (The lower bounds were environment switches until round 5; constants now.)
inline int env_int(const char* name, int fallback) {
    const char* e = std::getenv(name);
    return e ? std::atoi(e) : fallback;
}
constexpr int std_march_min_size() { return 31; }
constexpr int tpi_march_min_size() { return 1; }
// Disc size from which TPI on tiles with fractional elevations takes the two marching passes
// (sum of trunc(x), then sum of the fractional parts) instead of the general kernel.
// Disc size from which TPI takes the ring build (disc_ring_impl.hpp) instead of tpi_march_kernel (both give the same bits).
// Measured on the 32768^2 bench DEM (profiles/r02_tpi_ring.txt): 2.03-2.06 ms against
// 2.31-2.37 ms for 5 ... 11 px, level at 13-17 px, 4.6 ms against 4.4 ms at 67 px.  Round 3: the ring sizes go up
// to 17 px, because with fractional elevations the two-image pass (kRingBoth) halves the time there (7 px 5.36 ->
// 2.48 ms, 13 px 5.82 -> 3.60, 17 px 6.00 -> 3.79, profiles/r03_tpi_ring_both.txt) while whole metres cost the
// same to 15 px and 5 % more at 17 (2.43 against 2.32 ms).
constexpr int tpi_ring_min_size() { return 5; }
// Disc size from which STD / TPI + STD take the one-pass ring kernel.  Same-box
// A/B on the 32768^2 bench DEM (tools/std_time.py, profiles/r02_std_ring.txt), ring kernel against what it replaces:
// STD 7 px 3.10 / 5.67 ms, 17 px 3.77 / 6.40, 31 px 5.28 / 7.89, 45 px 6.76 / 9.70, 65 px 9.37 / 12.78, 67 px 10.33 /
// 13.61; TPI + STD 67 px 10.86 / 14.22.  Identical bits (CRC-32 of both planes, whole metres and fractional DEM).
constexpr int std_ring_min_size() { return 5; }
constexpr int tpi_ring_max_size() { return 17; }
// TOPO_AMD_TPI_FRACTION_EXACT=1: tiles with fractional elevations take the exact two-pass route (2^-16 m) instead of the
// scaled one-chain route (2^-8 m, tpi_scaled_march_kernel)
inline bool tpi_fraction_scaled() {
    static const int v = env_int("TOPO_AMD_TPI_FRACTION_EXACT", 0);
    return v == 0;
}
constexpr int tpi_fraction_min_size() { return 17; }
// (lab switch: TOPO_AMD_STD_SPEC_WIDE=0 keeps the 256-column strips at 5 and 7 px)
inline bool std_spec_wide() {
    static const int v = env_int("TOPO_AMD_STD_SPEC_WIDE", 1);
    return v != 0;
}

template <int SIZE>
int launch_wave_any(const Block& b, float* tpi_out, float* std_out) {
    constexpr int TH8 = tile_rows(SIZE, 8, 64), TH12 = tile_rows(SIZE, 12, 60);
    // The marching kernels take the tiles of whole metres, the general kernel the tiles they left
    // (none on a DEM of whole metres; all of them on one with fractional elevations).
    constexpr bool kFullTile = TH12 == 60;  // LDS holds the tile height the marching kernels were tuned for
    if constexpr (std_ring_fits(SIZE) && !std_ring_both_fits(SIZE) && kFullTile) {
        // STD (and TPI + STD) on a raster of mostly fractional elevations, discs of 43 ... 67 px (below, the ring kernel has
        // a three-image second pass; here three images do not fit LDS and every tile used to end with the general kernel's
        // three staging passes: 20 ms at 67 px on 32768^2).  Three marching passes instead, each carrying its window down
        // the strips: the sums of trunc(x) (and TPI of the whole-metre tiles), the sums of (trunc(x) - c)^2 (and STD of the
        // whole-metre tiles), the sums of the fractional parts with the finalisation of the rest.  Exact sums and the
        // general kernel's expressions: the same bits.  Which route runs first is a matter of time only; it is taken from
        // the share of fractional samples in the raster class.
        if (std_out && current_class().frac_share > 0.5f) {
            if (tpi_out) TOPO_TRY((launch_march_sums<SIZE, TH12, 12, true>(b, tpi_out)));
            else TOPO_TRY((launch_march_sums<SIZE, TH12, 12, false>(b, nullptr)));
            TOPO_TRY((launch_std_march<SIZE, TH12, 12, true>(b, std_out)));
            TOPO_TRY((launch_fraction_march<SIZE, TH12, 12, true>(b, tpi_out, std_out)));
            if (tpi_out) return launch_wave<SIZE, TH8, 8, true, true>(b, tpi_out, std_out, true, TH12);
            return launch_wave<SIZE, TH8, 8, false, true>(b, tpi_out, std_out, true, TH12);
        }
    }
    if constexpr (std_ring_fits(SIZE)) {
        if (std_out && SIZE >= std_ring_min_size()) {
            // one staging pass: u and u^2 rings side by side (disc_ring_impl.hpp), then the general kernel over
            // the tiles it marked (its map has this kernel's strips and rows of 60)
            // (the tiles at the DEM's border are the ring kernel's own: the padding's zeros are staged as samples, and such a
            // window is the general kernel's only when its highest sample passes 2 lim32 - disc_ring_impl.hpp)
            if constexpr (std_spec_wide_fits(SIZE)) {
                // A raster of whole metres (the raster class: TIME only - every route is exact): 512-column strips, 8 columns per
                // lane; a tile with fractional samples is the general kernel's then (no second pass on this map).  With row
                // segments of 2 KiB the two-plane call has no slow mode (tools/two_plane_pairs.py: 2.84 - 3.12 ms for all 30 pairs
                // of six planes at 7 px, where the 256-column strips run at 2.98 with one plane and 3.85 with the others), and
                // STD alone gains 2 - 6 % at 5 and 7 px.  From 9 px the ring holds one row per chain wave only and STD alone
                // loses (9 px 2.37 -> 3.11 ms) while TPI + STD still wins over its usual slow mode up to 13 px (3.60 -> 3.06,
                // 3.57 -> 3.23): profiles/r05_std_wide_ab.txt.
                // (STD alone on a small raster keeps the 256-column strips: 8192^2 has 18 strips of 480 columns for 17.07 and 12 tiles
                // per block where the narrow form, two blocks per CU, has 17.7: 0.148 against 0.161 ms)
                const bool big = (long)b.nx * b.out_rows >= (1L << 27);  // (a 4096-row shard of a 32768-column DEM included)
                if (current_class().frac_share == 0.0f && std_spec_wide() && (tpi_out != nullptr || (SIZE <= 7 && big))) {
                    if (tpi_out) TOPO_TRY((launch_std_ring_spec<SIZE, true, false, 8>(b, tpi_out, std_out)));
                    else TOPO_TRY((launch_std_ring_spec<SIZE, false, false, 8>(b, nullptr, std_out)));
                    if (tpi_out) return launch_wave<SIZE, TH8, 8, true, true>(b, tpi_out, std_out, true, StdRingCfg<SIZE>::TH, RGeo<SIZE, 8>::TILE_W);
                    return launch_wave<SIZE, TH8, 8, false, true>(b, tpi_out, std_out, true, StdRingCfg<SIZE>::TH, RGeo<SIZE, 8>::TILE_W);
                }
            }
            if constexpr (std_ring_spec(SIZE)) {  // the small discs: staging waves apart from chain waves
                if (tpi_out) TOPO_TRY((launch_std_ring_spec<SIZE, true>(b, tpi_out, std_out)));
                else TOPO_TRY((launch_std_ring_spec<SIZE, false>(b, nullptr, std_out)));
            } else {
                if (tpi_out) TOPO_TRY((launch_std_ring<SIZE, true>(b, tpi_out, std_out)));
                else TOPO_TRY((launch_std_ring<SIZE, false>(b, nullptr, std_out)));
            }
            if constexpr (std_ring_both_fits(SIZE)) {
                // tiles with fractional elevations: one more pass of the ring kernel with a third image (the
                // fractional parts) instead of the general kernel's three staging passes
                if constexpr (std_spec_both_fits(SIZE)) {  // (up to 21 px: in the form with staging waves apart from chain waves)
                    if (tpi_out) TOPO_TRY((launch_std_ring_spec<SIZE, true, true>(b, tpi_out, std_out)));
                    else TOPO_TRY((launch_std_ring_spec<SIZE, false, true>(b, nullptr, std_out)));
                } else {
                    if (tpi_out) TOPO_TRY((launch_std_ring<SIZE, true, kStdBoth>(b, tpi_out, std_out)));
                    else TOPO_TRY((launch_std_ring<SIZE, false, kStdBoth>(b, nullptr, std_out)));
                }
            }
            if (tpi_out) return launch_wave<SIZE, TH8, 8, true, true>(b, tpi_out, std_out, true, StdRingCfg<SIZE>::TH);
            return launch_wave<SIZE, TH8, 8, false, true>(b, tpi_out, std_out, true, StdRingCfg<SIZE>::TH);
        }
    }
    if (std_out && (SIZE < std_march_min_size() || !kFullTile)) {
        // small discs: one general kernel (two passes per tile, the second one out of L2) moves three
        // planes where the marching pair moves five, and the chains are too short to matter
        if (tpi_out) return launch_wave<SIZE, TH8, 8, true, true>(b, tpi_out, std_out);
        return launch_wave<SIZE, TH8, 8, false, true>(b, tpi_out, std_out);
    }
    if (std_out) {
        // sum x (and TPI) marching, then sum u^2 -> STD marching over the same runs, then the
        // 8-wave general kernel (tiles of TH8 rows) over whatever overlaps a marked tile
        if (tpi_out) {
            TOPO_TRY((launch_march<SIZE, TH12, 12, true, true>(b, tpi_out)));
        } else {
            TOPO_TRY((launch_march<SIZE, TH12, 12, false, true>(b, nullptr)));
        }
        TOPO_TRY((launch_std_march<SIZE, TH12, 12>(b, std_out)));
        if (tpi_out) return launch_wave<SIZE, TH8, 8, true, true>(b, tpi_out, std_out, true, TH12);
        return launch_wave<SIZE, TH8, 8, false, true>(b, tpi_out, std_out, true, TH12);
    }
    if (SIZE < tpi_march_min_size()) return launch_wave<SIZE, TH12, 12, true, false>(b, tpi_out, std_out);
    // TPI alone: whole-metre tiles are finished by the first marching kernel; tiles with fractional
    // elevations get their exact sum of trunc(x) there and the sum of the fractional parts in the
    // second; the general kernel takes what neither could (non-finite or absurd samples)
    // the ring build (disc_ring_impl.hpp) is compiled for the sizes it wins at and for the headline sizes (A/B)
    constexpr bool kRing = ring_both_fits(SIZE) && SIZE <= 17;
    if constexpr (kRing) {
        if (SIZE >= tpi_ring_min_size() && SIZE <= tpi_ring_max_size()) {
            using RC = RingCfg<SIZE, 8>;
            constexpr int map_tw = RGeo<SIZE, 8>::TILE_W;
            // whole-metre tiles in the first pass; tiles with fractional elevations in ONE second pass with two rings
            // (trunc(x) and the fractional parts); what neither could take in the general kernel.  On the 32768^2 bench
            // DEM with fractional elevations: 7 px 5.36 -> 2.48 ms, 17 px 5.31 -> 3.79 ms (profiles/r03_tpi_ring_both.txt)
            TOPO_TRY((launch_ring<SIZE, 8, kRingMark>(b, tpi_out)));
            TOPO_TRY((launch_ring<SIZE, 8, kRingBoth>(b, tpi_out)));
            return launch_wave<SIZE, TH12, 12, true, false>(b, tpi_out, std_out, true, RC::TH, map_tw);
        }
    }
    if (SIZE < tpi_fraction_min_size()) {
        // small discs: the general kernel's two passes over one tile beat two marching kernels
        TOPO_TRY((launch_march<SIZE, TH12, 12, true, false, false>(b, tpi_out)));
        return launch_wave<SIZE, TH12, 12, true, false>(b, tpi_out, std_out, true);
    }
    if (tpi_fraction_scaled()) {
        // fractional tiles: one chain on x in units of 2^-8 m (tpi_scaled_march_kernel: <= 1.95 mm, see there)
        // (what the last call on this block reported, or - a block seen for the first time, a host-buffer call - what the
        // raster class says: the scaled build takes every tile)
        if (dem_memo_mostly_fractional(b) || current_class().frac_share > 0.5f) {
            TOPO_TRY((launch_scaled_march<SIZE, TH12, 12, true>(b, tpi_out)));
            return launch_wave<SIZE, TH12, 12, true, false>(b, tpi_out, std_out, true);
        }
        TOPO_TRY((launch_march<SIZE, TH12, 12, true, true, true>(b, tpi_out, true)));
        TOPO_TRY((launch_scaled_march<SIZE, TH12, 12>(b, tpi_out)));
        return launch_wave<SIZE, TH12, 12, true, false>(b, tpi_out, std_out, true);
    }
    TOPO_TRY((launch_march<SIZE, TH12, 12, true, true, true>(b, tpi_out)));
    TOPO_TRY((launch_fraction_march<SIZE, TH12, 12>(b, tpi_out)));
    return launch_wave<SIZE, TH12, 12, true, false>(b, tpi_out, std_out, true);
}

}  // namespace

}  // namespace topo
