// K1 ring build: TPI with the window kept as a ring of prefix rows, 8 columns per lane.
//
// What the chain costs on gfx950 (tools/ubench/valu_mix.hip, profiles/r02_valu_mix_rate.txt): v_add_u32 /
// v_sub_u32 issue in 2 cycles per wave and SIMD, every DPP form (and v_add3_u32) in 4, and a block of 12
// waves runs like one of 16 (its waves do not spread evenly over the 4 SIMDs).  With 4 columns per lane
// (disc_wave_impl.hpp) a 67-px disc needs 18 lane hops of 4 partial sums for 4 outputs and only 46 of
// 64 lanes end with a full sum; a third of the VALU time is transport.  This build:
//
//   * gives a lane NCR = 8 adjacent columns: 10 hops of 8 partial sums for 8 outputs, 54 of 64 lanes
//     valid (432 of 512 staged columns), each hop fused into an add (v_add_u32_dpp);
//   * visits the runs of the disc from the rim inwards with TWO chains (one collecting the columns right
//     of the pixel with wave_shl, one those left of it with wave_shr), so that a run's column sums are
//     formed right before their first use and die after the last: ~120 live column sums instead of 168;
//   * keeps the prefix rows Q in a RING of R = SIZE + B rows of LDS (B = 8 output rows per phase, one
//     per wave).  A row is written once, where it stays until the window has passed it: no LDS -> LDS
//     move of the carried window, no fix-up pass, no full restaging;
//   * stages by COLUMN ownership: wave w owns staged columns 64 w .. 64 w + 63, one per lane, for the
//     whole strip.  Each phase it loads its segment of the next 8 DEM rows (8 dword loads, in flight
//     during the chain), and after the chain (barrier) truncates, classifies, continues its running
//     uint32 prefix (one register) and writes 8 x ds_write_b32 over the 8 oldest rows (barrier);
//   * takes the pixel's own value from the column sum of the outermost run (a single tap).
//
// Blocks are 8 waves (2 per SIMD, up to 256 VGPRs).  Tile list, XCD-contiguous runs and the
// finalisation are those of tpi_march_kernel, so the results have the same bits.  The per-tile state
// map (p.defer) has this build's own geometry (strips of 54 x 8 columns, tiles of 64 rows); the
// general kernel reads it through WaveArgs::map_th / map_tw.  MODE kRingFraction is the second pass
// over the tiles kRingMainFrac marked kNeedsFraction (the ring then holds the fractional parts in units
// of 2^-16 m).
#pragma once

namespace topo {

namespace {

#ifndef RING_PRIO_SWITCH_D
#define RING_PRIO_SWITCH_D 2
#endif
#ifndef RING_LEAD
#define RING_LEAD 3
#endif
#ifdef RING_NOSB
#define RING_SB()
#else
#define RING_SB() __builtin_amdgcn_sched_barrier(0)
#endif
#define DPP_WAVE_SHR1 0x138  // lane i takes lane i - 1; lane 0 takes 0 (bound_ctrl)

__device__ __forceinline__ uint32_t hop_up(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, DPP_WAVE_SHR1, 0xf, 0xf, true);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int SIZE, int NCR>
struct RGeo {
    static constexpr DiscTable<SIZE> T = make_disc_table<SIZE>();
    static_assert(T.centre == 0 && T.off_min == -T.off_max, "odd disc sizes only");
    static_assert(NCR % 4 == 0, "16-byte pieces");
    static constexpr int M = T.off_max;
    static constexpr int PARTS = NCR / 4;           // 16-byte pieces per lane and row
    static constexpr int DL = (M + NCR - 1) / NCR;  // lane hops per side
    static constexpr int NVL_ALL = 64 - 2 * DL;     // lanes that end up with full sums
    // Strips whose width is a whole number of 128-byte lines (32 columns): every row segment a wave stores is then made of
    // full, aligned lines - with the widest strip (496 columns = 1984 bytes at 7 px) every other strip starts and ends inside a
    // line that a neighbour strip, on another CU at another time, completes.  For the TPI rings (8 columns per lane) up to
    // 11 px, which wait for memory: 32768^2, 7 px 2.00 -> 1.81 ms, 11 px 2.01 -> 1.83, level from 13 px (profiles/r05_aligned_strips.txt); the STD
    // kernels (4 columns per lane) wait for their chains and lose the lanes given up (17 px 2.91 -> 3.20 ms), so they keep the
    // widest strips.  (-DRING_ALIGN_STRIPS=0: the widest strips everywhere, for the A/B.)
#ifndef RING_ALIGN_STRIPS
#define RING_ALIGN_STRIPS 1
#endif
    static constexpr int NVL_LINES = NCR * NVL_ALL / 32 * 32 / NCR;
    static constexpr int NVL = RING_ALIGN_STRIPS && NCR == 8 && SIZE <= 11 ? NVL_LINES : NVL_ALL;  // lanes whose sums are stored
    static constexpr int TILE_W = NCR * NVL;        // valid output columns per strip
    static constexpr int X0 = NCR * DL;             // staged column of the first valid output
    static constexpr int W = 64 * NCR;              // staged columns = dwords per ring row
    static_assert(NVL >= 16 && NVL <= NVL_ALL, "disc too wide for one wavefront");
    static constexpr int NR = T.num_runs;
    // The two-sided chain: step D (lane distance, DL .. 0) adds, for output sub-column t, the lane's own
    // columns s at offsets di = NCR D + s - t >= 0 (right chain) and -NCR D + s - t < 0 (left chain).
    struct Sched {
        int first_step[SIZE];  // per run: the largest step that uses it
        int order[SIZE];       // the runs in the order they are first needed (rim first)
        int centre_run;        // run with lo = hi = 0 (the pixel's own row only), or -1
    };
    static constexpr Sched make() {
        Sched t{};
        for (int r = 0; r < SIZE; ++r) t.first_step[r] = -1;
        for (int r = 0; r < SIZE; ++r) t.order[r] = 0;
        t.centre_run = -1;
        for (int D = 0; D <= DL; ++D)
            for (int s = 0; s < NCR; ++s)
                for (int q = 0; q < NCR; ++q) {
                    const int dr = NCR * D + s - q, dl = -NCR * D + s - q;
                    if (dr >= 0 && dr <= M) {
                        const int r = T.run_of[dr - T.off_min];
                        if (D > t.first_step[r]) t.first_step[r] = D;
                    }
                    if (dl < 0 && dl >= -M) {
                        const int r = T.run_of[dl - T.off_min];
                        if (D > t.first_step[r]) t.first_step[r] = D;
                    }
                }
        int n = 0;
        for (int D = DL; D >= 0; --D)
            for (int d = M; d >= 0; --d) {  // within a step: rim first
                const int r = T.run_of[d - T.off_min];
                bool seen = false;
                for (int i = 0; i < n; ++i) seen = seen || t.order[i] == r;
                if (!seen && t.first_step[r] == D) t.order[n++] = r;
            }
        for (int r = 0; r < NR; ++r)
            if (T.run_lo[r] == 0 && T.run_hi[r] == 0) t.centre_run = r;
        return t;
    }
    static constexpr Sched S = make();
    static_assert(S.centre_run >= 0, "the disc's outermost column is the pixel's own row");
};

// Ring row layout: PARTS pieces of 256 dwords; lane l keeps columns NCR l + 4 P .. + 3 at dwords
// 256 P + 4 l .. + 3, so that every ds_read_b128 of a wave is 1 KiB of consecutive bytes.
// min (MAX = false) or max of an int over the wavefront, the same value in every lane (wave-uniform)
template <bool MAX>
__device__ __forceinline__ int wave_min_max(int v) {
    auto op = [](int a, int b) { return MAX ? max(a, b) : min(a, b); };
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false));  // row_ror:1
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false));  // row_ror:2
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));  // row_ror:4
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));  // row_ror:8
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return op(op(r0, r1), op(r2, r3));
}

template <int NCR>
__device__ __forceinline__ int ring_dword_of_column(int c) {
    return ((c % NCR) / 4) * 256 + (c / NCR) * 4 + (c % 4);
}

// Disc sums of one output row.  ring: LDS image; s0: ring slot of the row's Q index 0 (prefix through the
// DEM row above its window), wave-uniform; acc[t]: sum over the disc for the lane's own column NCR lane + t,
// valid for DL <= lane < 64 - DL; ctr[t]: the staged value of the pixel itself.  The prefix rows of run
// i + LEAD are fetched before the column sums of run i are formed.
// P0 / P1 (CHAIN_PRIO, disc_wave_impl.hpp): issue priority of the wave for the first and the second half of the
// chain's steps (-1: leave it alone), so that the waves sharing a SIMD advance together instead of one after the other.
// SB: scheduling fences around the column sums of a run (the large discs: they keep the reads of run i + LEAD ahead of the
// subtractions of run i); the small discs' build leaves the scheduler free to interleave the chains of its rows.
template <int SIZE, int NCR, int R, int LEAD, int PITCH = 64 * NCR, int P0 = -1, int P1 = -1, bool SB = true>
__device__ __forceinline__ void ring_disc_sum(const uint32_t* ring, int s0, int lane, uint32_t (&acc)[NCR],
                                              uint32_t (&ctr)[NCR], bool young) {
    using G = RGeo<SIZE, NCR>;
    constexpr int NR = G::NR;
    constexpr int DL = G::DL;
    constexpr int M = G::M;
    u32x4 top[NR][G::PARTS], bot[NR][G::PARTS];
    uint32_t cv[NR][NCR];
    uint32_t aR[NCR], aL[NCR];
    const char* col = reinterpret_cast<const char*>(ring + lane * 4);
    // The slot of a prefix row is wave-uniform: its byte offset is formed on the scalar unit and reaches the address
    // with ONE full-rate v_add_u32 (sgpr + lane base); the pieces of a row are immediates.  (Left to the compiler
    // the slot index went through a v_lshl_add_u32 per read and the lower row of a pair through a further add of
    // a literal: three half- and full-rate vector instructions per run and chain, 126 per row of std_ring_kernel.)
    // byte offset of ring slot s0, and the wrap: (b0 + d) mod RB as min(b0 + d, b0 + d - RB) in unsigned arithmetic
    // (3 scalar instructions per row address)
    constexpr uint32_t RB = (uint32_t)R * PITCH * 4;
    const uint32_t b0 = (uint32_t)s0 * (PITCH * 4);
    auto fetch = [&](int i) {
        const int r = G::S.order[i];
        const uint32_t dt = (uint32_t)(G::T.run_hi[r] + 1 + M) * (PITCH * 4), db = (uint32_t)(G::T.run_lo[r] + M) * (PITCH * 4);
        const uint32_t ot = min(b0 + dt, b0 + dt - RB), ob = min(b0 + db, b0 + db - RB);
#pragma unroll
        for (int P = 0; P < G::PARTS; ++P) {
            top[r][P] = *reinterpret_cast<const u32x4*>(col + ot + P * 1024);
            bot[r][P] = *reinterpret_cast<const u32x4*>(col + ob + P * 1024);
        }
    };
#pragma unroll
    for (int i = 0; i < LEAD && i < NR; ++i) fetch(i);
    // Two waves share a SIMD and the hardware serves the older one first: waves 0-3 finish their chain in
    // 74 % of the time of waves 4-7 (s_memtime stamps, -DRING_STAMPS) and then idle at the barrier.  Giving
    // the younger wave priority for the first part of the chain (-DRING_PRIO) moved that around without
    // shortening the phase (4.61 ms against 4.62 ms at 67 px), so it is off.
#ifdef RING_PRIO
    if (young) __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int D = DL; D >= 0; --D) {
#ifdef RING_PRIO
        if (D == RING_PRIO_SWITCH_D && young) __builtin_amdgcn_s_setprio(0);
#endif
        if (P1 >= 0 && D == DL / 2) CHAIN_SETPRIO(P1);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            if (G::S.first_step[G::S.order[j]] == D) {
                const int r = G::S.order[j];
                if (j + LEAD < NR) fetch(j + LEAD);
                if (SB) RING_SB();
#pragma unroll
                for (int P = 0; P < G::PARTS; ++P)
#pragma unroll
                    for (int s = 0; s < 4; ++s) cv[r][4 * P + s] = top[r][P][s] - bot[r][P][s];
                if (SB) RING_SB();
            }
        }
#pragma unroll
        for (int t = 0; t < NCR; ++t) {
            uint32_t pr = 0, pl = 0;
            bool anyr = false, anyl = false;
#pragma unroll
            for (int s = 0; s < NCR; ++s) {
                const int dr = NCR * D + s - t;
                const int dl = -NCR * D + s - t;
                if (dr >= 0 && dr <= M) {
                    const uint32_t c = cv[G::T.run_of[dr - G::T.off_min]][s];
                    pr = anyr ? pr + c : c;
                    anyr = true;
                }
                if (dl < 0 && dl >= -M) {
                    const uint32_t c = cv[G::T.run_of[dl - G::T.off_min]][s];
                    pl = anyl ? pl + c : c;
                    anyl = true;
                }
            }
            if (D == DL) {
                aR[t] = pr;
                aL[t] = pl;
            } else {
                // the lane's own part is made opaque so that the hop's add stays a two-operand add and the
                // DPP move folds into it (v_add_u32_dpp, 4 cycles) instead of feeding a v_add3_u32 (4 + 4)
                if (anyr) {
                    asm("" : "+v"(pr));
                    aR[t] = hop(aR[t]) + pr;
                } else {
                    aR[t] = hop(aR[t]);
                }
                if (anyl) {
                    asm("" : "+v"(pl));
                    aL[t] = hop_up(aL[t]) + pl;
                } else {
                    aL[t] = hop_up(aL[t]);
                }
            }
        }
        if (SB) RING_SB();
    }
#pragma unroll
    for (int t = 0; t < NCR; ++t) {
        acc[t] = aR[t] + aL[t];
        ctr[t] = cv[G::S.centre_run][t];
    }
}

template <int SIZE, int NCR>
struct RingCfg {
    using G = RGeo<SIZE, NCR>;
    static constexpr int NW = 8;        // waves per block = output rows per phase = rows staged per phase
    static constexpr int B = NW;
    static constexpr int TH = 64;       // rows of a map tile
    static constexpr int PPT = TH / B;  // phases per tile
    static constexpr int R = SIZE + B;  // ring slots: the window of a phase (Q indices 0 .. SIZE of B rows)
    static constexpr int HALO = SIZE - 1;
    // stream rows above the first window row: >= 1 (Q index 0 = "through the row above"), and such that the
    // rows staged before phase 0 are whole batches
    static constexpr int PAD = 1 + (B - (1 + HALO + B) % B) % B;
    static constexpr int PRO = PAD + HALO + B;               // rows staged before phase 0
    static constexpr int NB_PRO = PRO / B;                   // as batches
    static constexpr int HIST = (SIZE + 2 * B - 1) / B + 1;  // batches a phase's window can touch (one to spare)
    static_assert(G::W / 64 == NW, "one staged 64-column segment per wave");
    static constexpr size_t LDS = (size_t)R * G::W * sizeof(uint32_t) + 2 * NW * sizeof(int) + 16;
    static_assert(TH % B == 0 && PRO % B == 0, "whole batches");
    static_assert(LDS <= 160 * 1024, "ring does not fit LDS");
    static_assert(HIST <= 31, "flag history");
};

constexpr bool ring_fits(int size, int ncr) {
    return size >= 5 && (size_t)(size + 8) * 64 * ncr * 4 + 128 <= 160 * 1024 && 64 - 2 * ((size / 2 + ncr - 1) / ncr) >= 16;
}

enum RingMode { kRingMain = 0, kRingMainFrac = 1, kRingFraction = 2, kRingMark = 3, kRingBoth = 4 };

// the two-image pass (kRingBoth) fits LDS and has its strips?  (5 ... 31 px with 8 columns per lane)
constexpr bool ring_both_fits(int size) {
    return size >= 5 && size % 2 == 1 && (size_t)(size + 8) * 64 * 8 * 4 * 2 + 128 <= 160 * 1024;
}

// kRingMain: tiles with fractional samples are left to the general kernel.  kRingMainFrac: their exact sums of
// trunc(x) go to p.sums and they are marked kNeedsFraction.  kRingFraction: the second pass over those tiles
// (sum of the fractional parts, then TPI with the expression and operands of tpi_fraction_march_kernel).
// kRingMark + kRingBoth: the pair for small discs on DEMs with fractional elevations.  kRingMark is kRingMain except
// that a tile with fractional samples is marked kNeedsFraction (and not computed) instead of going to the general
// kernel; kRingBoth then takes those tiles in ONE pass with two rings side by side - the prefix rows of trunc(x) and of
// the fractional parts in units of 2^-16 m - runs both chains and finalises with the expression and operands of
// kRingFraction (sum of trunc(x) as an exact integer, plus 2^-16 x the integer sum of the parts), hence with the bits
// of the fraction passes and of the general kernel.  One read of the DEM and one write of TPI instead of the sums
// plane going out and coming back between two passes.
// SIZE2 (0 = none): a second, smaller disc evaluated from the same ring in the same pass (several
// scales from one read of the DEM).  The window of the smaller disc is inside the larger one's, so the staging, the
// ring and the tile classification are those of SIZE; its chain runs over its own run table with the ring slot of its
// own first row, and TPI of SIZE2 goes to p.tpi2 with the expression of the single call, hence with its bits.  A tile
// the pass leaves to the general kernel is left for both sizes (kRingMain only: discs below 17 px).
template <int SIZE, int NCR, int MODE, int SIZE2 = 0>
__device__ __forceinline__ void tpi_ring_kernel_body(const WaveArgs& p, int tiles_x, int tiles_y, const PartRun deal, const int vb0, const int nb) {
    static_assert(SIZE2 == 0 || (SIZE2 < SIZE && SIZE2 >= 5 && SIZE2 % 2 == 1 && MODE == kRingMain), "pair: a smaller odd disc, main pass");
    using G = RGeo<SIZE, NCR>;
    using C = RingCfg<SIZE, NCR>;
    constexpr int B = C::B, R = C::R, PPT = C::PPT, NW = C::NW;
    constexpr int DL = G::DL;
    constexpr bool FRACTION = MODE == kRingFraction;
    constexpr bool BOTH = MODE == kRingBoth;
    constexpr bool SECOND = FRACTION || BOTH;  // a second pass: works on the tiles the map names
    static_assert(!BOTH || SIZE2 == 0, "the two-image pass takes one disc");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    uint32_t* Q = lds_u;
    uint32_t* F = Q + R * G::W;  // kRingBoth: the ring of the fractional parts
    int* wflags = reinterpret_cast<int*>(Q + (BOTH ? 2 : 1) * R * G::W);  // [2 parities][NW]: what each wave saw in the batch it staged

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ntiles = tiles_x * tiles_y;
    const int vb = (vb0 + deal.shift) % nb;
    // the block's contiguous run of the strip-major tile list (deal_parts)
    const int first = deal.first(vb);
    const int last = min(first + deal.count(vb), ntiles);
    const double inv_nm1 = 1.0 / ((double)G::T.taps - 1.0);

    if (SECOND) {
        // nothing to do on a DEM of whole metres: one flag per lane, 64 tiles per load
        bool any = false;
        for (int base = first; base < last; base += 64) {
            const int mine = base + lane;
            any = any || __builtin_amdgcn_ballot_w64(mine < last && p.defer[mine < last ? mine : first] == kNeedsFraction) != 0;
        }
        if (!any) return;  // the same for every thread of the block
    }

    // staging: this lane's column of this wave's segment
    const int scol = 64 * wave + lane;                 // staged column
    const int sdw = ring_dword_of_column<NCR>(scol);   // its dword in a ring row
    const int rmin = max(0, p.in_row0), rmax = min(p.gny, p.in_row0 + p.in_rows);  // DEM rows this block may read

#pragma unroll 1
    for (int tile0 = first; tile0 < last;) {
        // one run: the block's tiles of one column strip, top to bottom
        const int ty0 = tile0 % tiles_y;
        const int strip = tile0 / tiles_y;
        const int run_tiles = min(last - tile0, tiles_y - ty0);
        const int nphase = run_tiles * PPT;
        const int ox0 = strip * G::TILE_W;
        const int oyS = (p.out_row0 / C::TH + ty0) * C::TH;  // global multiples of TH
        const int gx0 = ox0 - G::X0;                          // global column of staged column 0
        // stream row n is DEM row gy0 + n; its prefix row goes to ring slot n % R, so Q index k of output row
        // oyS + j sits in slot (j + k + PAD - 1) % R
        const int gy0 = oyS - G::M - C::PAD;
        const int gcol = gx0 + scol;
        const bool col_ok = gcol >= 0 && gcol < p.nx;
        const float* src = p.in + (col_ok ? gcol : 0);
        auto load_batch = [&](int n0, float (&v)[B]) {
#pragma unroll
            for (int r = 0; r < B; ++r) {
                int gy = gy0 + n0 + r;
                gy = min(max(gy, rmin), rmax - 1);  // a clamped row is read and thrown away
                v[r] = src[(size_t)(gy - p.in_row0) * p.nx];
            }
        };
        if (MODE == kRingMain || MODE == kRingMark) {
            // A tile with fractional elevations is not for this pass (kRingMain: the general kernel's; kRingMark:
            // the two-image pass's): a run whose four probed rows all hold fractional or non-finite samples is
            // handed over unstaged (see std_ring_kernel)
            int odd = 0;
            const int pc = min(max(ox0 + 4 * lane, 0), p.nx - 1);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int gy = oyS + ((2 * u + 1) * run_tiles * C::TH) / 8;
                gy = min(max(gy, rmin), rmax - 1);
                const float x = p.in[(size_t)(gy - p.in_row0) * p.nx + pc];
                odd += __builtin_amdgcn_ballot_w64(!(x == truncf(x))) != 0 ? 1 : 0;
            }
            if (odd == 4) {
                for (int t = tile0 + (int)threadIdx.x; t < tile0 + run_tiles; t += NW * 64)
                    p.defer[t] = MODE == kRingMark ? kNeedsFraction : kTileGeneral;
                tile0 += run_tiles;
                continue;
            }
        }
        if (SECOND) {
            // a run without a tile for this pass is not staged (a DEM that is fractional in places only)
            bool any = false;
            for (int base = tile0; base < tile0 + run_tiles; base += 64) {
                const int mine = base + lane;
                any = any || __builtin_amdgcn_ballot_w64(mine < tile0 + run_tiles && p.defer[mine < tile0 + run_tiles ? mine : tile0] == kNeedsFraction) != 0;
            }
            if (!any) {
                tile0 += run_tiles;
                continue;
            }
        }
        uint32_t run = 0;    // running prefix of this lane's column (wraps, harmlessly)
        uint32_t run_f = 0;  // kRingBoth: the same for the fractional parts
        int wslot = 0;     // ring slot of the next row to stage
        // Staging a batch of B rows in two halves.  convert_batch: the loaded samples -> the prefix values
        // to write (registers) and what the wave saw; it ends with the wave's only wait on its loads, so it
        // runs before any younger store is issued (vmcnt counts loads and stores together, in order, and a
        // wait for the last load behind two output stores would wait for the stores too: 1300 cycles per
        // phase in the first version).  write_batch: the B ds_write_b32 over the B oldest rows.
        auto convert_batch = [&](int n0, const float (&v)[B], uint32_t (&q)[B], uint32_t (&qf)[BOTH ? B : 1]) {
            uint32_t amax = 0;  // largest |x| seen, as float bits (NaN / inf sort above all)
            bool frac = false;
#pragma unroll
            for (int r = 0; r < B; ++r) {
                const int gy = gy0 + n0 + r;
                const bool ok = col_ok && gy >= rmin && gy < rmax;
                const float x = ok ? v[r] : 0.0f;  // padding is staged as zero (mode="same")
                if (FRACTION) {
                    run += stage_value<kStF>(x, 0.0f, 0);
                } else {
                    // (int)x truncates towards zero; NaN -> 0 and |x| >= 2^31 saturates, both caught by amax
                    const int xi = (int)x;
                    frac |= x != (float)xi;
                    amax = max(amax, __float_as_uint(x) & 0x7fffffffu);
                    run += (uint32_t)xi;
                    if (BOTH) {
                        run_f += stage_value<kStF>(x, 0.0f, 0);
                        qf[r] = run_f;
                    }
                }
                q[r] = run;
            }
            int wf = 0;
            if (!FRACTION) {
                // |trunc(x)| <= kAbsLim  <=>  |x| < kAbsLim + 1
                if (__builtin_amdgcn_ballot_w64(frac)) wf |= kTileFrac;
                if (__builtin_amdgcn_ballot_w64(amax >= __float_as_uint(kAbsLim + 1.0f))) wf |= kTileFloat;
            }
            return wf;
        };
        auto write_batch = [&](const uint32_t (&q)[B], const uint32_t (&qf)[BOTH ? B : 1]) {
#pragma unroll
            for (int r = 0; r < B; ++r) {
                int s = wslot + r;
                s = s >= R ? s - R : s;
                Q[s * G::W + sdw] = q[r];
                if (BOTH) F[s * G::W + sdw] = qf[r];
            }
            wslot += B;
            wslot = wslot >= R ? wslot - R : wslot;
        };
        auto stage_batch = [&](int n0, const float (&v)[B]) {
            uint32_t q[B], qf[BOTH ? B : 1];
            const int wf = convert_batch(n0, v, q, qf);
            write_batch(q, qf);
            return wf;
        };
        unsigned hist_frac = 0, hist_float = 0;
        // every wave folds the 8 waves' flags of the batch just staged into its (identical) history
        auto fold_flags = [&](int parity) {
            const int4 a = *reinterpret_cast<const int4*>(wflags + parity * NW);
            const int4 b = *reinterpret_cast<const int4*>(wflags + parity * NW + 4);
            const int all = __builtin_amdgcn_readfirstlane(a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w);
            hist_frac = ((hist_frac << 1) | ((all & kTileFrac) ? 1u : 0u)) & ((1u << C::HIST) - 1u);
            hist_float = ((hist_float << 1) | ((all & kTileFloat) ? 1u : 0u)) & ((1u << C::HIST) - 1u);
        };

        // ---- prologue: the window of phase 0, two batches of loads in flight; leaves the batch that follows
        // the prologue in va ----
        float va[B], vb2[B];
        int pro_flags = 0;
        load_batch(0, va);
#pragma unroll 1
        for (int k = 0; k < C::NB_PRO; k += 2) {
            load_batch((k + 1) * B, vb2);
            pro_flags |= stage_batch(k * B, va);
            if (k + 1 < C::NB_PRO) {
                load_batch((k + 2) * B, va);
                pro_flags |= stage_batch((k + 1) * B, vb2);
            }
        }
        if (C::NB_PRO % 2 != 0) {
#pragma unroll
            for (int r = 0; r < B; ++r) va[r] = vb2[r];
        }
        if (lane == 0) wflags[wave] = pro_flags;
        __syncthreads();
        fold_flags(0);

        const int ocol = gx0 + lane * NCR;  // the lane's own columns
        const bool lane_ok = lane >= DL && lane < DL + G::NVL;
        int s0 = C::PAD - 1 + wave;  // slot of Q index 0 of this wave's row in phase 0
        int mode = kTileDone;
#ifdef RING_STAMPS
        long long tsum[5] = {0, 0, 0, 0, 0};
#define RING_STAMP(i) { const long long now_ = __builtin_amdgcn_s_memtime(); tsum[i] += now_ - tlast; tlast = now_; }
        long long tlast = __builtin_amdgcn_s_memtime();
#else
#define RING_STAMP(i)
#endif
#pragma unroll 1
        for (int ph = 0; ph < nphase; ++ph) {
            const int tile = tile0 + ph / PPT;
            if (SECOND) {
                if (ph % PPT == 0) mode = __builtin_amdgcn_readfirstlane((int)p.defer[tile]);
                if (BOTH && hist_float && mode == kNeedsFraction) {
                    // (a run the first pass handed over unstaged was not classified there) non-finite or absurd
                    // samples in the window: the whole tile is the general kernel's
                    mode = kTileGeneral;
                    if (threadIdx.x == 0) p.defer[tile] = (uint8_t)kTileGeneral;
                }
            } else {
                constexpr bool kFracLater = MODE == kRingMainFrac || MODE == kRingMark;  // fractional tiles have a pass of their own
                const int now = hist_float ? kTileGeneral : (hist_frac ? (kFracLater ? kNeedsFraction : kTileGeneral) : kTileDone);
                if (ph % PPT == 0) {
                    mode = now;
                    if (threadIdx.x == 0) p.defer[tile] = (uint8_t)mode;
                } else if (MODE == kRingMark && mode == kTileDone && now == kNeedsFraction) {
                    // fractional rows further down the tile: all of it goes to the two-image pass (the rows written
                    // so far are written again there, with the same bits)
                    mode = kNeedsFraction;
                    if (threadIdx.x == 0) p.defer[tile] = (uint8_t)kNeedsFraction;
                } else if (mode != kTileGeneral && now != mode && !(mode == kNeedsFraction && now == kTileDone)) {
                    // the tile's later rows brought in what its first phase had not seen: leave all of it
                    mode = kTileGeneral;
                    if (threadIdx.x == 0) p.defer[tile] = (uint8_t)kTileGeneral;
                }
            }
            const bool compute = SECOND ? mode == kNeedsFraction : (MODE == kRingMark ? mode == kTileDone : mode != kTileGeneral);
            uint32_t next_q[B], next_qf[BOTH ? B : 1];
            int next_flags = 0;
            bool converted = false;
            if (compute) {
                const int oy = oyS + ph * B + wave;
                const bool live = lane_ok && oy >= p.out_row0 && oy < p.out_row0 + p.out_rows && ocol < p.nx;
                const bool live2 = live && ocol + 4 < p.nx;  // nx % 4 == 0: the lane's second 16 bytes may lie outside
                const size_t o = live ? (size_t)(oy - p.out_row0) * p.nx + ocol : 0;
                Vec4<int> sv[2];
                Vec4<float> xs[2];
                if (SECOND) {
                    const size_t xi = live ? (size_t)(oy - p.in_row0) * p.nx + ocol : 0;
#pragma unroll
                    for (int P = 0; P < 2; ++P) {
                        const bool lv = P == 0 ? live : live2;
                        if (FRACTION) sv[P] = *reinterpret_cast<const Vec4<int>*>(p.sums + (lv ? o + 4 * P : 0));
                        xs[P] = *reinterpret_cast<const Vec4<float>*>(p.in + (lv ? xi + 4 * P : 0));
                    }
                }
                uint32_t acc[NCR], ctr[NCR];
                if constexpr (SIZE2 != 0) {
                    // the smaller disc first (its sums are converted and stored before the larger chain needs the registers)
                    using G2 = RGeo<SIZE2, NCR>;
                    constexpr int kShift = G::M - G2::M;  // its window starts this many rows further down
                    int s2 = s0 + kShift;
                    s2 = s2 >= R ? s2 - R : s2;
                    uint32_t acc2[NCR], ctr2[NCR];
                    ring_disc_sum<SIZE2, NCR, R, RING_LEAD>(Q, s2, lane, acc2, ctr2, wave >= NW / 2);
                    const double inv_nm1_2 = 1.0 / ((double)G2::T.taps - 1.0);
#pragma unroll
                    for (int P = 0; P < 2; ++P) {
                        if (P == 0 ? live : live2) {
                            Vec4<float> out_t;
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const int xi = (int)ctr2[4 * P + t];
                                out_t.v[t] = (float)((double)xi - (double)((int)acc2[4 * P + t] - xi) * inv_nm1_2);
                            }
                            *reinterpret_cast<Vec4<float>*>(p.tpi2 + o + 4 * P) = out_t;
                        }
                    }
                }
                ring_disc_sum<SIZE, NCR, R, RING_LEAD>(Q, s0, lane, acc, ctr, wave >= NW / 2);
                if (BOTH) {
                    // the sum of trunc(x) takes the place of the first pass's plane of sums; then the fractional parts
#pragma unroll
                    for (int t = 0; t < NCR; ++t) sv[t / 4].v[t % 4] = (int)acc[t];
                    ring_disc_sum<SIZE, NCR, R, RING_LEAD>(F, s0, lane, acc, ctr, wave >= NW / 2);
                }
                if (ph + 1 < nphase) next_flags = convert_batch(C::PRO + ph * B, va, next_q, next_qf);
                converted = true;
                static_assert(NCR == 8, "the stores below write two 16-byte pieces per lane");
#pragma unroll
                for (int P = 0; P < 2; ++P) {
                    if (P == 0 ? live : live2) {
                        if (MODE == kRingMainFrac && mode == kNeedsFraction) {
                            const Vec4<int> s4{{(int)acc[4 * P], (int)acc[4 * P + 1], (int)acc[4 * P + 2], (int)acc[4 * P + 3]}};
                            *reinterpret_cast<Vec4<int>*>(p.sums + o + 4 * P) = s4;
                        } else {
                            Vec4<float> out_t;
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                if (SECOND) {
                                    const double sf = (double)(int)acc[4 * P + t] * (1.0 / 65536.0);
                                    const double s1 = (double)sv[P].v[t] + sf;
                                    const double x_ctr = (double)xs[P].v[t];
                                    out_t.v[t] = (float)((double)xs[P].v[t] - (s1 - x_ctr) * inv_nm1);
                                } else {
                                    const int xi = (int)ctr[4 * P + t];  // integers: see tpi_march_kernel
                                    out_t.v[t] = (float)((double)xi - (double)((int)acc[4 * P + t] - xi) * inv_nm1);
                                }
                            }
                            *reinterpret_cast<Vec4<float>*>(p.tpi + o + 4 * P) = out_t;
                        }
                    }
                }
            }
            if (SECOND && ph % PPT == PPT - 1 && mode == kNeedsFraction && threadIdx.x == 0) p.defer[tile] = kTileDone;
            if (!converted && ph + 1 < nphase) next_flags = convert_batch(C::PRO + ph * B, va, next_q, next_qf);
            s0 += B;
            s0 = s0 >= R ? s0 - R : s0;
            RING_STAMP(0)
            __syncthreads();  // every wave is done with the B oldest rows
            RING_STAMP(1)
            if (ph + 1 < nphase) {
                write_batch(next_q, next_qf);
                load_batch(C::PRO + (ph + 1) * B, va);  // in flight during the next chain
                if (lane == 0) wflags[((ph + 1) & 1) * NW + wave] = next_flags;
            }
            RING_STAMP(2)
            __syncthreads();
            RING_STAMP(3)
            if (ph + 1 < nphase) fold_flags((ph + 1) & 1);
        }
#ifdef RING_STAMPS
        if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 131) && MODE != kRingFraction)
            printf("blk %d wave %d phases %d: chain %lld  barA %lld  stage %lld  barB %lld (memtime ticks)\n", (int)blockIdx.x, wave,
                   nphase, tsum[0], tsum[1], tsum[2], tsum[3]);
#endif
        tile0 += run_tiles;
    }
}

template <int SIZE, int NCR, int MODE, int SIZE2 = 0>
__global__ __launch_bounds__(512) void tpi_ring_kernel(WaveArgs p, int tiles_x, int tiles_y, PartRun deal) {
    TOPO_RUN_ONE((tpi_ring_kernel_body<SIZE, NCR, MODE, SIZE2>));
}
template <int SIZE, int NCR, int MODE, int SIZE2 = 0>
__global__ __launch_bounds__(512) void tpi_ring_kernel_parts(WaveParts ps, int tiles_x) {
    TOPO_RUN_PARTS((tpi_ring_kernel_body<SIZE, NCR, MODE, SIZE2>));
}

template <int SIZE, int NCR, int MODE, int SIZE2 = 0>
int launch_ring(const Block& b, float* tpi_out, float* tpi2_out = nullptr) {
    using G = RGeo<SIZE, NCR>;
    using C = RingCfg<SIZE, NCR>;
    Context& c = ctx();
    WaveArgs a{b.in, tpi_out, nullptr, b.in_rows, b.in_row0, b.gny, b.nx, b.out_row0, b.out_rows,
               nullptr, nullptr, nullptr, 0, 0, 0, tpi2_out};
    constexpr size_t kLds = C::LDS + (MODE == kRingBoth ? (size_t)C::R * G::W * sizeof(uint32_t) : 0);
    static_assert(kLds <= 160 * 1024, "the two rings do not fit LDS");
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        TOPO_HIP(hipFuncSetAttribute((const void*)tpi_ring_kernel<SIZE, NCR, MODE, SIZE2>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
        int nblk = 0;
        TOPO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)tpi_ring_kernel<SIZE, NCR, MODE, SIZE2>,
                                                              C::NW * 64, kLds));
        blocks_per_cu = nblk < 1 ? 1 : (nblk > 4 ? 4 : nblk);  // small discs: several rings per CU
    }
    WaveParts ps;
    int tiles_x = 0;
    long ntiles = 0;
    TOPO_TRY(make_parts(b, a, C::TH, G::TILE_W, true, MODE == kRingMainFrac || MODE == kRingFraction, &ps, &tiles_x, &ntiles));
    const long grid = march_grid(c, blocks_per_cu, ntiles);
    deal_parts(&ps, tiles_x, grid, blocks_per_cu);
    return launch_parts(tpi_ring_kernel<SIZE, NCR, MODE, SIZE2>, tpi_ring_kernel_parts<SIZE, NCR, MODE, SIZE2>, grid, C::NW * 64, kLds, ps, tiles_x);
}


// ---- K2 ring build: STD and TPI + STD from ONE staging pass ------------------------------------------------
// The marching pair (tpi_march_kernel<OUT_SUM> + std_march_kernel) reads the DEM twice, writes and re-reads a
// plane of sums and runs the general kernel over what is left: 27 GB of traffic for 8.6 GB of work at 67 px.
// Here the prefix rows of u = trunc(x) - c AND of u^2 sit side by side in one ring (2 KiB per row, 4 columns
// per lane: two images of 8 columns per lane do not fit), both chains of an output row run in the wave that
// owns the row, and STD (and TPI) are finalised from registers with the expressions of the other kernels:
//   STD = std_from_int_sums(sum u, sum u^2): shift-invariant, so the bits do not depend on c;
//   TPI = x - (sum trunc(x) - x) / (n - 1) with sum trunc(x) = sum u + c n, an exact integer.
// 12 waves: every wave takes one row per phase (B = 12, 5 phases per 60-row map tile); waves 0-3 also own the
// staging of 64 columns each.
//
// The offset c.  Everything is uint32 arithmetic modulo 2^32; the only true constraint is that the sum of u^2
// over a disc stays below 2^32, i.e. |u| <= lim32 for the samples of the windows that are evaluated.  The
// stagers keep, per batch of rows, the smallest and largest trunc(x) they saw; at a tile boundary the block
// looks at the range of the rows now in the ring and ahead of it and, when c has drifted from its middle,
// RE-BASES THE RING IN PLACE: with delta = c' - c, a prefix row of k rows transforms as
//   Q1' = Q1 - delta k,   Q2' = Q2 - 2 delta Q1 + delta^2 k        (exact modulo 2^32; additive constants
// in k or Q1 cancel in the differences the chains take), 79 rows x 512 dwords in about a microsecond, no reload.
// A window whose relief exceeds 2 lim32 (2244 m at 67 px) cannot be centred: its tile goes to the general
// kernel through the map (rows of 60, this kernel's own strips), like the tiles with fractional or non-finite
// samples and the tiles whose discs leave the DEM (their in-domain tap counts differ from n).
// The small discs take the build with staging waves apart from chain waves (std_ring_spec_kernel, below): its batches are 16
// rows, so the map tiles of those sizes - this kernel's second pass (kStdBoth) works on the same map - are 48 rows.
// (kStdSpecMax, std_tile_rows: common.hpp - the shard launcher aligns its seams to these tile rows)
constexpr bool std_ring_spec(int size) { return size >= 5 && size % 2 == 1 && size <= kStdSpecMax; }
constexpr int std_ring_tile_rows(int size) { return std_tile_rows(size); }

template <int SIZE>
struct StdRingCfg {
    using G = RGeo<SIZE, 4>;
    static constexpr int NW = 12;
    static constexpr int B = NW;
    static constexpr int TH = std_ring_tile_rows(SIZE);
    static constexpr int PPT = TH / B;
    static constexpr int R = SIZE + B;
    static constexpr int PITCH = 2 * G::W;  // dwords per ring row: the u image, then the u^2 image
    static constexpr int HALO = SIZE - 1;
    static constexpr int PAD = 1 + (B - (1 + HALO + B) % B) % B;
    static constexpr int PRO = PAD + HALO + B;
    static constexpr int NB_PRO = PRO / B;
    static constexpr int HIST = (SIZE + 2 * B - 1) / B + 1;
    static constexpr int SW = G::W / 64;  // staging waves
    static constexpr size_t LDS = (size_t)R * PITCH * sizeof(uint32_t) + 2 * NW * 4 * sizeof(int) + 16;
    static_assert(TH % B == 0 && PRO % B == 0, "whole batches");
    static_assert(LDS <= 160 * 1024, "ring does not fit LDS");
    static_assert(HIST <= 16, "flag history");
};

constexpr bool std_ring_fits(int size) {
    return size >= 5 && size % 2 == 1 && (size_t)(size + 12) * 512 * 4 + 512 <= 160 * 1024 && 64 - 2 * ((size / 2 + 3) / 4) >= 16;
}
// three images (u, u^2 and the fractional parts) fit: 5 ... 41 px
constexpr bool std_ring_both_fits(int size) {
    return std_ring_fits(size) && (size_t)(size + 12) * 768 * 4 + 512 <= 160 * 1024;
}

// kStdMain: what the comment above describes (tiles with fractional samples are left to what follows).
// kStdBoth, the second pass for DEMs with fractional elevations (discs up to 41 px): it takes the tiles kStdMain left
// (those at the DEM's border too: the padding's zeros are samples, below) with a THIRD image in the
// ring, the prefix rows of the fractional parts in units of 2^-16 m, runs the three chains and finalises every pixel
// with the general kernel's choice and expressions (the integer form when the window's fractional sum is exactly 0,
// else s1 = (sum u + c n) + 2^-16 sum g, s2 = sum u^2 + 2 c sum u + c^2 n in float64: exact integers whatever c is),
// hence with its bits.  A tile it cannot take either (non-finite samples, more relief than the u^2 sums hold) stays
// marked for the general kernel.  One read of the DEM instead of the general kernel's three staging passes:
// 32768^2 with fractional elevations, STD 7 px 6.97 -> 4.10 ms, 17 px 8.34 -> 5.18, 31 px 10.88 -> 7.99, 41 px
// 12.56 -> 10.81 (profiles/r03_std_ring_both.txt).
// The tiles at the DEM's border (round 5).  A tap outside the DEM reads 0 in the reference (zero padding, mode="same"), so
// it is staged as what it is - a SAMPLE of elevation 0, u = -c - and not as "no tap": then sum u and sum u^2 run over all n
// taps, T = Su + c n and S2 = Su2 + 2 c Su + c^2 n like anywhere else, and the finalisation is the interior one (n S2 - T^2:
// the exact integer std_from_border_sums forms from the in-domain sums and the in-domain count m, hence the same bits).  No tap
// counts, no second finalisation, nothing added to the phase loop; the zeros simply belong to the window's range, so the offset
// follows to about half the terrain's height and a border window fits while its highest sample stays below 2 lim32 (18.7 km at
// 7 px, 4.9 km at 31 px, 2.3 km at 65 px); a tile that does not fit is the general kernel's, like a window with too much relief
// anywhere.  Before: rounds 2 - 4 left every border tile to the general kernel (a trailing launch over 3 % of the pixels at an
// eighth of the rate on an under-filled grid: 20 % of an 8192^2 step at 7 px); earlier in round 5 they were a second launch of
// this kernel over a host-built list with the tap counts in the finalisation (43 us of a 269 us step: two tiles per CU one
// after the other; inside the phase loop that finalisation cost 50 - 60 spilled registers).  Rows outside the block's VIEW
// (a row block's missing neighbours) are not samples: they are staged at u = 0, keep the range as it is and reach no output.
enum StdRingMode { kStdMain = 0, kStdBoth = 2 };

template <int SIZE, bool WANT_TPI, int MODE = kStdMain>
__device__ __forceinline__ void std_ring_kernel_body(const WaveArgs& p, int tiles_x, int tiles_y, const PartRun deal, const int vb0, const int nb) {
    using G = RGeo<SIZE, 4>;
    using C = StdRingCfg<SIZE>;
    constexpr bool BOTH = MODE == kStdBoth;
    // the tiles at the DEM's border are computed here (above) up to 39 px: beyond, the windows of Alpine terrain are too wide
    // with the padding's zeros in them (2 lim32 = 3.6 km at 41 px - where the bench DEM's border tiles fail half way down - 2.3 km at 65 px), and the kernels of the large discs sit at
    // the register limit - whatever their phase loop gains in scalar work it pays in spills (same box, 32768^2, STD 65 px:
    // 7.91 ms with the border tiles left to the general kernel at once, 8.33 - 8.68 ms with the test per tile)
    constexpr bool kBorderHere = SIZE <= 39;
    constexpr int B = C::B, R = C::R, PPT = C::PPT, NW = C::NW, HIST = C::HIST;
    constexpr int PITCH = (BOTH ? 3 : 2) * G::W;  // dwords per ring row: the u image, the u^2 image (and the image of the fractional parts)
    constexpr int DL = G::DL;
    constexpr int kBig = 0x3fffffff;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    uint32_t* Q = lds_u;
    int* wflags = reinterpret_cast<int*>(Q + R * PITCH);  // [2 parities][NW][4]: flags, min, max of the batch a wave staged
    // more than 128 registers: the block's 12 waves then cannot sit 4 + 4 + 2 + 2 on the SIMDs
    asm volatile("" ::: "v140");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ntiles = tiles_x * tiles_y;
    const int vb = (vb0 + deal.shift) % nb;
    // the block's contiguous run of the strip-major tile list (deal_parts)
    const int first = deal.first(vb);
    const int last = min(first + deal.count(vb), ntiles);
    const double n = (double)G::T.taps;
    const double inv_n = 1.0 / n;
    const double inv_nm1 = 1.0 / (n - 1.0);
    const double inv_nn1 = 1.0 / (n * (n - 1.0));
    const int lim32 = (int)floorf(sqrtf(4294967295.0f / (float)G::T.taps));
    const int rmin = max(0, p.in_row0), rmax = min(p.gny, p.in_row0 + p.in_rows);
    const bool stager = wave < C::SW;
    // kStdBoth: a tile the first pass left for its fractional samples (41 px: with every disc of its pixels inside the DEM)
    auto candidate = [&](int t) {
        if (!kBorderHere) {
            const int ty = t % tiles_y, strip = t / tiles_y;
            const int oy0 = (p.out_row0 / C::TH + ty) * C::TH, ox0 = strip * G::TILE_W;
            if (!(oy0 - G::M >= 0 && oy0 + C::TH - 1 + G::M <= p.gny - 1 && ox0 - G::M >= 0 && ox0 + G::TILE_W - 1 + G::M <= p.nx - 1)) return false;
        }
        return p.defer[t] == kTileGeneral;
    };
    if (BOTH) {
        // nothing to do on a DEM of whole metres: one tile per lane, 64 tiles per ballot
        bool any = false;
        for (int base = first; base < last; base += 64) {
            const int mine = base + lane;
            any = any || __builtin_amdgcn_ballot_w64(mine < last && candidate(mine < last ? mine : first)) != 0;
        }
        if (!any) return;  // the same for every thread of the block
    }
    const int scol = 64 * wave + lane;  // staged column of a staging lane
    int seen_general = 0;

#pragma unroll 1
    for (int pos = first; pos < last;) {
        // a run: consecutive tiles of one strip
        const int tile0 = pos;
        const int run_tiles = min(last - tile0, tiles_y - tile0 % tiles_y);
        const int ty0 = tile0 % tiles_y;
        const int strip = tile0 / tiles_y;
        const int nphase = run_tiles * PPT;
        const int ox0 = strip * G::TILE_W;
        const int oyS = (p.out_row0 / C::TH + ty0) * C::TH;
        const int gx0 = ox0 - G::X0;
        const int gy0 = oyS - G::M - C::PAD;
        // the run starts around the sample at the centre of its first tile; it is re-based as the data drifts
        int cy = min(max(oyS + C::TH / 2, 0), p.gny - 1);
        cy = min(max(cy, p.in_row0), p.in_row0 + p.in_rows - 1);
        const int cx = min(ox0 + G::TILE_W / 2, p.nx - 1);
        float cf = truncf(p.in[(size_t)(cy - p.in_row0) * p.nx + cx]);
        if (!(fabsf(cf) <= kAbsLim)) cf = 0.0f;
        // (a run that starts at the DEM's border has the padding's zeros in its first windows: half way)
        if (kBorderHere && (ox0 - G::M < 0 || ox0 + G::TILE_W - 1 + G::M > p.nx - 1 || oyS - G::M < 0 || oyS + C::TH - 1 + G::M > p.gny - 1))
            cf = truncf(0.5f * cf);
        // (the large discs: every disc of the strip's pixels stays inside the DEM's columns?)
        const bool cols_inside = ox0 - G::M >= 0 && ox0 + G::TILE_W - 1 + G::M <= p.nx - 1;
        int ci = __builtin_amdgcn_readfirstlane((int)cf);
        const int gcol = gx0 + scol;
        const bool col_ok = stager && gcol >= 0 && gcol < p.nx;
        const float* src = p.in + (col_ok ? gcol : 0);
        auto load_batch = [&](int n0, float (&v)[B]) {
            if (!stager) return;
#pragma unroll
            for (int r = 0; r < B; ++r) {
                int gy = gy0 + n0 + r;
                gy = min(max(gy, rmin), rmax - 1);
                v[r] = src[(size_t)(gy - p.in_row0) * p.nx];
            }
        };
        if (BOTH) {
            // a run without a tile for this pass is not staged
            bool any = false;
            for (int base = tile0; base < tile0 + run_tiles; base += 64) {
                const int mine = base + lane;
                any = any || __builtin_amdgcn_ballot_w64(mine < tile0 + run_tiles && candidate(mine < tile0 + run_tiles ? mine : tile0)) != 0;
            }
            if (!any) {
                pos += run_tiles;
                continue;
            }
        } else {
            // On a DEM with fractional elevations every tile ends up with the general kernel, and staging the
            // strip for nothing cost 3 ms at 67 px (20.4 -> 23.5 ms).  Four rows spread over the run are probed
            // first (64 columns each, the same in every wave, so the block agrees): when all four hold a
            // fractional or non-finite sample the run is handed over unstaged.  A run that is only partly
            // fractional and slips through is still exact: its tiles are marked one by one below.
            int odd = 0;
            const int pc = min(max(ox0 + 4 * lane, 0), p.nx - 1);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int gy = oyS + ((2 * u + 1) * run_tiles * C::TH) / 8;
                gy = min(max(gy, rmin), rmax - 1);
                const float x = p.in[(size_t)(gy - p.in_row0) * p.nx + pc];
                odd += __builtin_amdgcn_ballot_w64(!(x == truncf(x))) != 0 ? 1 : 0;
            }
            if (odd == 4) {
                for (int t = tile0 + (int)threadIdx.x; t < tile0 + run_tiles; t += NW * 64) p.defer[t] = kTileGeneral;
                pos += run_tiles;
                continue;
            }
        }
        uint32_t run_u = 0, run_u2 = 0, run_f = 0;
        int wslot = 0;  // ring slot of the next row to stage = slot of the oldest row
        // what a stager saw in a batch: flags, smallest and largest trunc(x)
        struct Seen { int flags, lo, hi; };
        auto convert_batch = [&](int n0, const float (&v)[B], uint32_t (&q)[B], uint32_t (&q2)[B], uint32_t (&qf)[BOTH ? B : 1]) {
            Seen s{0, kBig, -kBig};
            if (!stager) return s;
            uint32_t amax = 0;  // largest |x| as float bits (NaN / inf sort above all)
            bool frac = false;
            int lo = kBig, hi = -kBig;
            // outside the DEM: a sample of elevation 0, the reference's padding; outside the block's view: nothing (u = 0).
            // Two copies of the loop, chosen by the batch (the same for the whole block): rows inside the view need one select
            // per sample - the four stagers are what a phase of the small discs waits for.
            auto rows = [&](auto inside_tag) {
                constexpr bool INSIDE = decltype(inside_tag)::value;  // every row of the batch is inside the block's view
#pragma unroll
                for (int r = 0; r < B; ++r) {
                    const int gy = gy0 + n0 + r;
                    const bool ok = INSIDE ? col_ok : (col_ok && gy >= rmin && gy < rmax);
                    float x;
                    if (INSIDE) {
                        x = ok ? v[r] : 0.0f;
                    } else {
                        // (the large discs leave the border tiles alone: whatever is staged outside the DEM reaches no output)
                        const bool pad = kBorderHere && (!col_ok || gy < 0 || gy >= p.gny);
                        x = ok ? v[r] : (pad ? 0.0f : (float)ci);
                    }
                    const int t = (int)x;                  // truncation towards zero; NaN -> 0, caught by amax
                    frac |= x != (float)t;
                    amax = max(amax, __float_as_uint(x) & 0x7fffffffu);
                    lo = min(lo, t);
                    hi = max(hi, t);
                    const uint32_t u = (uint32_t)(t - ci);
                    run_u += u;
                    // (24-bit multiply, full rate: a sample that passes the classification has |trunc(x)| <= 2^18 and the
                    // offset follows the data, so |u| < 2^23; what a flagged sample leaves in the ring is never used)
                    run_u2 += (uint32_t)__mul24((int)u, (int)u);
                    q[r] = run_u;
                    q2[r] = run_u2;
                    if (BOTH) {
                        run_f += stage_value<kStF>(ok ? x : 0.0f, 0.0f, 0);
                        qf[r] = run_f;
                    }
                }
            };
            if (kBorderHere && gy0 + n0 >= rmin && gy0 + n0 + B <= rmax) rows(std::true_type{});
            else rows(std::false_type{});
            // wave-wide min / max: four rotations inside the rows of 16 lanes (DPP), then the four rows through the
            // scalar unit.  (Twelve __shfl_xor, i.e. LDS round trips one after the other, sat on the staging waves'
            // path to the barrier every phase.)
            lo = wave_min_max<false>(lo);
            hi = wave_min_max<true>(hi);
            if (__builtin_amdgcn_ballot_w64(frac)) s.flags |= kTileFrac;
            if (__builtin_amdgcn_ballot_w64(amax >= __float_as_uint(kAbsLim + 1.0f))) s.flags |= kTileFloat;
            s.lo = lo;
            s.hi = hi;
            return s;
        };
        auto write_batch = [&](const uint32_t (&q)[B], const uint32_t (&q2)[B], const uint32_t (&qf)[BOTH ? B : 1]) {
            if (stager) {
#pragma unroll
                for (int r = 0; r < B; ++r) {
                    int sl = wslot + r;
                    sl = sl >= R ? sl - R : sl;
                    Q[sl * PITCH + scol] = q[r];
                    Q[sl * PITCH + G::W + scol] = q2[r];
                    if (BOTH) Q[sl * PITCH + 2 * G::W + scol] = qf[r];
                }
            }
            wslot += B;
            wslot = wslot >= R ? wslot - R : wslot;
        };
        auto publish = [&](int parity, const Seen& s) {
            if (lane == 0 && stager) {
                int* w = wflags + (parity * NW + wave) * 4;
                w[0] = s.flags;
                w[1] = s.lo;
                w[2] = s.hi;
            }
        };
        unsigned hist = 0;           // per batch: fractional / non-finite / absurd samples seen (kStdBoth: the last two only)
        int hlo[HIST], hhi[HIST];    // per batch: range of trunc(x) (index 0 = the newest batch)
#pragma unroll
        for (int k = 0; k < HIST; ++k) hlo[k] = kBig, hhi[k] = -kBig;
        auto fold = [&](int parity) {
            int fl = 0, lo = kBig, hi = -kBig;
#pragma unroll
            for (int w = 0; w < C::SW; ++w) {
                const int* q = wflags + (parity * NW + w) * 4;
                fl |= q[0];
                lo = min(lo, q[1]);
                hi = max(hi, q[2]);
            }
            fl = __builtin_amdgcn_readfirstlane(fl);
            constexpr int kStops = BOTH ? kTileFloat : (kTileFrac | kTileFloat);
            hist = ((hist << 1) | ((fl & kStops) ? 1u : 0u)) & ((1u << HIST) - 1u);
#pragma unroll
            for (int k = HIST - 1; k > 0; --k) hlo[k] = hlo[k - 1], hhi[k] = hhi[k - 1];
            hlo[0] = __builtin_amdgcn_readfirstlane(lo);
            hhi[0] = __builtin_amdgcn_readfirstlane(hi);
        };

        // prologue: the window of phase 0, one batch at a time (a run starts a handful of times per launch: the
        // exposed load latency does not count, the registers of a second buffer would)
        float va[B];
        Seen pro{0, kBig, -kBig};
#pragma unroll 1
        for (int k = 0; k < C::NB_PRO; ++k) {
            uint32_t q[B], q2[B], qf[BOTH ? B : 1];
            load_batch(k * B, va);
            const Seen s = convert_batch(k * B, va, q, q2, qf);
            pro.flags |= s.flags;
            pro.lo = min(pro.lo, s.lo);
            pro.hi = max(pro.hi, s.hi);
            write_batch(q, q2, qf);
        }
        load_batch(C::PRO, va);
        publish(0, pro);
        __syncthreads();
        fold(0);

        const int ocol = gx0 + lane * 4;
        const bool lane_ok = lane >= DL && lane < DL + G::NVL;
        int s0 = C::PAD - 1 + wave;
        int tmode = kTileDone;  // what the map says about the current tile
#ifdef STD_STAMPS  // (lab build, tools/ubench/tpi_lab.hip: s_memtime around the parts of a phase, summed per wave)
        long long tsum[6] = {0, 0, 0, 0, 0, 0};
#define STD_STAMP(i) { const long long now_ = __builtin_amdgcn_s_memtime(); tsum[i] += now_ - tlast; tlast = now_; }
        long long tlast = __builtin_amdgcn_s_memtime();
#else
#define STD_STAMP(i)
#endif
#pragma unroll 1
        for (int ph = 0; ph < nphase; ++ph) {
            const int tile = tile0 + ph / PPT;
            int wlo = kBig, whi = -kBig;  // range of the rows the windows of this phase (and a bit more) can touch
#pragma unroll
            for (int k = 0; k < HIST; ++k) wlo = min(wlo, hlo[k]), whi = max(whi, hhi[k]);
            if (ph % PPT == 0) {
                // a new tile: re-base the ring when c has drifted from the middle of the range
                // (or when the window does not fit around c where it is and would around its middle: the border windows, whose
                // range starts at the padding's 0, are often close to the 2 lim32 the chain of squares takes).  A tile at the
                // DEM's border has the padding's zeros in its windows, now or a few phases on: they belong to its range from
                // the start, and when that range is too wide (Alpine terrain under a 65-px disc) the tile is the general
                // kernel's at once, as every border tile was before - not after some of its rows have been computed for nothing.
                const int oy0 = oyS + (ph / PPT) * C::TH;
                bool at_border, too_wide, rebase;
                int mid;
                if (kBorderHere) {
                    at_border = ox0 - G::M < 0 || ox0 + G::TILE_W - 1 + G::M > p.nx - 1 || oy0 - G::M < 0 || oy0 + C::TH - 1 + G::M > p.gny - 1;
                    const int tlo = at_border ? min(wlo, 0) : wlo, thi = at_border ? max(whi, 0) : whi;
                    too_wide = at_border && tlo <= thi && thi - tlo > 2 * lim32;
                    mid = tlo + (thi - tlo) / 2;
                    const bool fits_now = thi - ci <= lim32 && ci - tlo <= lim32;
                    rebase = tlo <= thi && !too_wide && (abs(mid - ci) > lim32 / 4 || (!fits_now && thi - tlo <= 2 * lim32));
                } else {
                    at_border = !cols_inside || oy0 - G::M < 0 || oy0 + C::TH - 1 + G::M > p.gny - 1;
                    too_wide = at_border;  // (the general kernel's)
                    mid = wlo + (whi - wlo) / 2;
                    rebase = wlo <= whi && abs(mid - ci) > lim32 / 4;
                }
                if (rebase) {
                    const uint32_t delta = (uint32_t)(mid - ci);
                    const uint32_t d2 = delta * delta;
                    // k: the rows of the column up to this one, counted from the oldest row in the ring (any constant added to
                    // it cancels in the differences the chains take).  Every row moves with c - the padding's zeros are samples
                    // like the others; what the rows outside a block's view hold reaches no output.
                    for (int idx = threadIdx.x; idx < R * G::W; idx += NW * 64) {
                        const int sl = idx / G::W, col = idx - sl * G::W;
                        int since = sl - wslot;  // rows since the oldest one in the ring
                        since = since < 0 ? since + R : since;
                        const uint32_t k = (uint32_t)since + 1u;
                        uint32_t* q = Q + sl * PITCH + col;
                        const uint32_t q1 = q[0];
                        q[G::W] = q[G::W] - 2u * delta * q1 + d2 * k;
                        q[0] = q1 - delta * k;  // (the fractional parts do not depend on c)
                    }
                    if (stager) {  // (through the newest row: k = R)
                        run_u2 = run_u2 - 2u * delta * run_u + d2 * (uint32_t)R;
                        run_u = run_u - delta * (uint32_t)R;
                    }
                    ci = mid;
                    __syncthreads();
                }
                if (BOTH) {
                    tmode = candidate(tile) ? kNeedsFraction : kTileDone;  // (kTileDone here: not this pass's)
                } else {
                    tmode = too_wide ? kTileGeneralWide : kTileDone;
                    if (threadIdx.x == 0) p.defer[tile] = (uint8_t)tmode;
                }
            }
            // the windows of this phase hold only finite samples within lim32 of c (kStdMain: and whole ones)?
            const bool fits = hist == 0 && whi - ci <= lim32 && ci - wlo <= lim32;
            if (BOTH) {
                if (tmode == kNeedsFraction && !fits) tmode = kTileGeneral;  // the mark stays: general kernel
            } else if ((tmode & 1) == 0 && !fits) {
                tmode = hist != 0 ? kTileGeneral : kTileGeneralWide;  // (fractional samples: the second pass may take it)
                if (threadIdx.x == 0) p.defer[tile] = (uint8_t)tmode;
            }
            const bool compute = BOTH ? tmode == kNeedsFraction : tmode == kTileDone;
            uint32_t nq[B], nq2[B], nqf[BOTH ? B : 1];
            if (compute) {
                uint32_t su[4], ctr[4], su2[4], dummy[4];
// progress-based issue priority (CHAIN_PRIO, disc_wave_impl.hpp) from 25 px: 67 px 10.19 -> 9.37 ms, 31 px
                // 5.31 -> 5.03 ms; at 7 px the chains are too short for it (3.10 -> 3.28 ms)
                constexpr int kP = SIZE >= 25 ? 0 : -4;
                ring_disc_sum<SIZE, 4, R, 2, PITCH, 3 + kP, 2 + kP>(Q, s0, lane, su, ctr, false);
                ring_disc_sum<SIZE, 4, R, 2, PITCH, 1 + kP, 0 + kP>(Q + G::W, s0, lane, su2, dummy, false);
                const int oy = oyS + ph * B + wave;
                if (BOTH) {
                    uint32_t sfi[4];
                    ring_disc_sum<SIZE, 4, R, 2, PITCH>(Q + 2 * G::W, s0, lane, sfi, dummy, false);
                    if (lane_ok && oy >= p.out_row0 && oy < p.out_row0 + p.out_rows && ocol < p.nx) {
                        const size_t o = (size_t)(oy - p.out_row0) * p.nx + ocol;
                        Vec4<float> xs{{0.f, 0.f, 0.f, 0.f}};
                        if (WANT_TPI) xs = *reinterpret_cast<const Vec4<float>*>(p.in + (size_t)(oy - p.in_row0) * p.nx + ocol);
                        Vec4<float> out_s, out_t;
                        const double cd = (double)ci;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            // the general kernel's expressions (disc_wave_kernel, finalise_row) on the same exact sums
                            const double dsu = (double)(int)su[t], dsu2 = (double)su2[t];
                            const double sf = (double)(int)sfi[t] * (1.0 / 65536.0);
                            const double s1 = (dsu + cd * n) + sf;
                            if (sfi[t] == 0) {
                                out_s.v[t] = std_from_int_sums((int)su[t], (uint64_t)su2[t], (uint32_t)G::T.taps, (float)inv_nn1);
                            } else {
                                const double s2 = dsu2 + 2.0 * cd * dsu + cd * cd * n;
                                out_s.v[t] = std_from_sums(s1, s2, inv_n, inv_nm1);
                            }
                            if (WANT_TPI) out_t.v[t] = (float)((double)xs.v[t] - (s1 - (double)xs.v[t]) * inv_nm1);
                        }
                        *reinterpret_cast<Vec4<float>*>(p.sd + o) = out_s;
                        if (WANT_TPI) *reinterpret_cast<Vec4<float>*>(p.tpi + o) = out_t;
                    }
                } else if (lane_ok && oy >= p.out_row0 && oy < p.out_row0 + p.out_rows && ocol < p.nx) {
                    const size_t o = (size_t)(oy - p.out_row0) * p.nx + ocol;
                    Vec4<float> out_s, out_t;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        out_s.v[t] = std_from_int_sums((int)su[t], (uint64_t)su2[t], (uint32_t)G::T.taps, (float)inv_nn1);
                        if (WANT_TPI) {
                            const int xi = (int)ctr[t] + ci;  // integers: see tpi_march_kernel
                            // sum of trunc(x) = su + c n: exact, fits int32
                            out_t.v[t] = (float)((double)xi - (double)((int)su[t] + ci * G::T.taps - xi) * inv_nm1);
                        }
                    }
                    *reinterpret_cast<Vec4<float>*>(p.sd + o) = out_s;
                    if (WANT_TPI) *reinterpret_cast<Vec4<float>*>(p.tpi + o) = out_t;
                }
            }
            if (BOTH && ph % PPT == PPT - 1 && tmode == kNeedsFraction && threadIdx.x == 0) p.defer[tile] = kTileDone;
            if (!BOTH && ph % PPT == PPT - 1 && tmode == kTileGeneral) ++seen_general;
            STD_STAMP(0)
            const Seen seen = convert_batch(C::PRO + ph * B, va, nq, nq2, nqf);
            s0 += B;
            s0 = s0 >= R ? s0 - R : s0;
            STD_STAMP(1)
            __syncthreads();
            STD_STAMP(2)
            write_batch(nq, nq2, nqf);
            load_batch(C::PRO + (ph + 1) * B, va);
            publish((ph + 1) & 1, seen);
            STD_STAMP(3)
            __syncthreads();
            STD_STAMP(4)
            fold((ph + 1) & 1);
            STD_STAMP(5)
        }
#ifdef STD_STAMPS
        if (lane == 0 && (blockIdx.x == 3 || blockIdx.x == 131) && (wave == 0 || wave == 5 || wave == 11) && pos == first)
            printf("blk %3d wave %2d phases %d: chain+finalise %lld | convert %lld | barrier %lld | write+loads %lld | barrier %lld | fold %lld   (memtime ticks, 100 MHz)\n",
                   (int)blockIdx.x, wave, nphase, tsum[0], tsum[1], tsum[2], tsum[3], tsum[4], tsum[5]);
#endif
        pos += run_tiles;
    }
    // what this block's run looked like, for the next call on this DEM (dem_memo, common.hpp): tiles, and tiles it left to
    // the general kernel (fractional elevations, mostly)
    if (!BOTH && p.report != nullptr && vb == nb / 2 && threadIdx.x == 0) {
        __hip_atomic_store(p.report + 1, (uint32_t)seen_general, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(p.report, (uint32_t)(last > first ? last - first : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int SIZE, bool WANT_TPI, int MODE = kStdMain>
__global__ __launch_bounds__(768) void std_ring_kernel(WaveArgs p, int tiles_x, int tiles_y, PartRun deal) {
    TOPO_RUN_ONE((std_ring_kernel_body<SIZE, WANT_TPI, MODE>));
}
template <int SIZE, bool WANT_TPI, int MODE = kStdMain>
__global__ __launch_bounds__(768) void std_ring_kernel_parts(WaveParts ps, int tiles_x) {
    TOPO_RUN_PARTS((std_ring_kernel_body<SIZE, WANT_TPI, MODE>));
}


// ---- The small discs: staging waves apart from chain waves (round 5) ------------------------------------------------
// s_memtime stamps of std_ring_kernel<7> (profiles/r05_std7_stamps.txt): a phase - 12 rows - takes 4900 cycles of which the
// chip's vector ALUs have work for 1500.  The rest is the order of things: the four staging waves run their chains, THEN
// convert the next batch (1400 cycles alone on their SIMDs: one wave issues an instruction every 4 - 5 cycles) while the other
// eight wait at the barrier, then write it and issue the loads (1200 cycles) behind a second barrier.  Here the two jobs
// run side by side: waves 0-3 only stage (convert batch ph + 1, write it into ring slots nobody reads yet, issue the loads of
// batch ph + 2), waves 4-11 only run chains (two output rows each per phase, 16 rows per phase), ONE barrier per phase.
// The ring has 16 more slots for that (R = SIZE + 32); everything else - the offset c and its re-basing, the classification
// per batch, the border's zeros as samples, the finalisation - is std_ring_kernel's, expression for expression, so the bits are.
// Used for 5 ... 41 px (kStdSpecMax): same-box A/B on the 32768^2 bench DEM (tools/std_time.py, profiles/r05_std_spec_ab.txt), STD
// 7 px 2.67 -> 2.33 ms, 17 px 3.35 -> 2.91, 25 px 4.20 -> 3.83, 31 px 4.58 -> 4.43, 41 px 5.98 -> 5.40; level at 43 - 47 px (where
// eight chain waves carry what twelve did), and from 49 px the ring has no room for the second batch.
// NCR = 8 (the whole-metre pass of the smallest discs): 512-column strips like the TPI rings' - eight staging waves, eight chain
// waves with two rows of 8 columns per lane each, a block of 16 waves
template <int SIZE, bool BOTH = false, int NCR = 4>
struct StdSpecCfg {
    using G = RGeo<SIZE, NCR>;
    // (512-column strips: two rows per chain wave while two images of SIZE + 32 rows of 2 KiB fit - 5 and 7 px - one beyond)
    static constexpr int SW = G::W / 64, CW = 8, NW = SW + CW;
    static constexpr int RPW = NCR == 8 && (size_t)(SIZE + 32) * 1024 * 4 + 1024 > 160 * 1024 ? 1 : 2;
    static constexpr int B = CW * RPW;  // rows per phase and per batch: 16
    static constexpr int TH = std_ring_tile_rows(SIZE);
    static constexpr int PPT = TH / B;
    static constexpr int R = SIZE + 2 * B;
    static constexpr int PITCH = (BOTH ? 3 : 2) * G::W;  // the u image, the u^2 image (and the image of the fractional parts)
    static constexpr int HALO = SIZE - 1;
    static constexpr int PAD = 1 + (B - (1 + HALO + B) % B) % B;
    static constexpr int PRO = PAD + HALO + B;
    static constexpr int NB_PRO = PRO / B;
    static constexpr int HIST = (SIZE + 2 * B - 1) / B + 1;
    static constexpr size_t LDS = (size_t)R * PITCH * sizeof(uint32_t) + 2 * NW * 4 * sizeof(int) + 16;
    // 5 and 7 px: two rings fit a CU's LDS, and the kernel is short of waves (every wave runs a serial stream of dependent
    // instructions): two blocks per CU, 6 waves per SIMD, 84 registers
    static constexpr bool kTwoBlocks = NCR == 4 && 2 * LDS <= 160 * 1024;
    static constexpr int kWavesPerSimd = NCR == 8 ? 4 : (kTwoBlocks ? 6 : 3);
    static_assert((SW == 4 || SW == 8) && (B == 16 || B == 8), "four or eight staging waves, eight chain waves with one or two rows each");
    static_assert(NCR == 4 || !BOTH, "three images of 512 columns do not fit");
    static_assert(TH % B == 0 && PRO % B == 0, "whole batches");
    static_assert(LDS <= 160 * 1024, "ring does not fit LDS");
    static_assert(HIST <= 16, "flag history");
};
// the second pass for fractional elevations (std_ring_kernel's kStdBoth: a third image, the three chains, the general
// kernel's expressions) in this form: the three images of SIZE + 32 rows fit up to 21 px
// the 512-column build: two images of SIZE + 32 rows of 2 KiB
// (used for 5 and 7 px, and for TPI + STD up to 13 px: launch_wave_any)
constexpr bool std_spec_wide_fits(int size) { return std_ring_spec(size) && size <= 13; }
constexpr bool std_spec_both_fits(int size) { return std_ring_spec(size) && (size_t)(size + 32) * 768 * 4 + 512 <= 160 * 1024; }

template <int SIZE, bool WANT_TPI, bool BOTH = false, int NCR = 4>
__device__ __forceinline__ void std_ring_spec_body(const WaveArgs& p, int tiles_x, int tiles_y, const PartRun deal, const int vb0, const int nb) {
    using G = RGeo<SIZE, NCR>;
    using C = StdSpecCfg<SIZE, BOTH, NCR>;
    constexpr int B = C::B, R = C::R, PPT = C::PPT, NW = C::NW, HIST = C::HIST, PITCH = C::PITCH;
    constexpr int DL = G::DL;
    constexpr int kBig = 0x3fffffff;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    uint32_t* Q = lds_u;
    int* wflags = reinterpret_cast<int*>(Q + R * PITCH);  // [2 parities][NW][4]: flags, min, max of the batch a wave staged
    // more than 128 registers: the block's 12 waves then cannot sit 4 + 4 + 2 + 2 on the SIMDs (the two-blocks-per-CU build of
    // the smallest discs: 84 at most)
    if (!C::kTwoBlocks && NCR == 4) asm volatile("" ::: "v140");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ntiles = tiles_x * tiles_y;
    const int vb = (vb0 + deal.shift) % nb;
    const int first = deal.first(vb);
    const int last = min(first + deal.count(vb), ntiles);
    const double n = (double)G::T.taps;
    const double inv_n = 1.0 / n;
    const double inv_nm1 = 1.0 / (n - 1.0);
    const double inv_nn1 = 1.0 / (n * (n - 1.0));
    const int lim32 = (int)floorf(sqrtf(4294967295.0f / (float)G::T.taps));
    const int rmin = max(0, p.in_row0), rmax = min(p.gny, p.in_row0 + p.in_rows);
    const bool stager = wave < C::SW;
    const int scol = 64 * wave + lane;  // staged column of a staging lane
    const int sdw = ring_dword_of_column<NCR>(scol);  // ... and its dword in a ring row (RGeo: 16-byte pieces per lane)
    int seen_general = 0;
    // BOTH: a tile the first pass left for its fractional samples
    auto candidate = [&](int t) { return p.defer[t] == kTileGeneral; };
    if (BOTH) {
        // nothing to do on a DEM of whole metres: one tile per lane, 64 tiles per ballot
        bool any = false;
        for (int base = first; base < last; base += 64) {
            const int mine = base + lane;
            any = any || __builtin_amdgcn_ballot_w64(mine < last && candidate(mine < last ? mine : first)) != 0;
        }
        if (!any) return;  // the same for every thread of the block
    }

#pragma unroll 1
    for (int pos = first; pos < last;) {
        const int tile0 = pos;
        const int run_tiles = min(last - tile0, tiles_y - tile0 % tiles_y);
        const int ty0 = tile0 % tiles_y;
        const int strip = tile0 / tiles_y;
        const int nphase = run_tiles * PPT;
        const int ox0 = strip * G::TILE_W;
        const int oyS = (p.out_row0 / C::TH + ty0) * C::TH;
        const int gx0 = ox0 - G::X0;
        const int gy0 = oyS - G::M - C::PAD;
        int cy = min(max(oyS + C::TH / 2, 0), p.gny - 1);
        cy = min(max(cy, p.in_row0), p.in_row0 + p.in_rows - 1);
        const int cx = min(ox0 + G::TILE_W / 2, p.nx - 1);
        float cf = truncf(p.in[(size_t)(cy - p.in_row0) * p.nx + cx]);
        if (!(fabsf(cf) <= kAbsLim)) cf = 0.0f;
        const bool cols_border = ox0 - G::M < 0 || ox0 + G::TILE_W - 1 + G::M > p.nx - 1;
        if (cols_border || oyS - G::M < 0 || oyS + C::TH - 1 + G::M > p.gny - 1) cf = truncf(0.5f * cf);
        int ci = __builtin_amdgcn_readfirstlane((int)cf);
        const int gcol = gx0 + scol;
        const bool col_ok = stager && gcol >= 0 && gcol < p.nx;
        const float* src = p.in + (col_ok ? gcol : 0);
        auto load_batch = [&](int n0, float (&v)[B]) {
#pragma unroll
            for (int r = 0; r < B; ++r) {
                int gy = gy0 + n0 + r;
                gy = min(max(gy, rmin), rmax - 1);
                v[r] = src[(size_t)(gy - p.in_row0) * p.nx];
            }
        };
        if (BOTH) {
            bool any = false;
            for (int base = tile0; base < tile0 + run_tiles; base += 64) {
                const int mine = base + lane;
                any = any || __builtin_amdgcn_ballot_w64(mine < tile0 + run_tiles && candidate(mine < tile0 + run_tiles ? mine : tile0)) != 0;
            }
            if (!any) {  // a run without a tile for this pass is not staged
                pos += run_tiles;
                continue;
            }
        } else {
            // a run of fractional elevations is handed over unstaged (std_ring_kernel)
            int odd = 0;
            const int pc = min(max(ox0 + 4 * lane, 0), p.nx - 1);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int gy = oyS + ((2 * u + 1) * run_tiles * C::TH) / 8;
                gy = min(max(gy, rmin), rmax - 1);
                const float x = p.in[(size_t)(gy - p.in_row0) * p.nx + pc];
                odd += __builtin_amdgcn_ballot_w64(!(x == truncf(x))) != 0 ? 1 : 0;
            }
            if (odd == 4) {
                for (int t = tile0 + (int)threadIdx.x; t < tile0 + run_tiles; t += NW * 64) p.defer[t] = kTileGeneral;
                pos += run_tiles;
                continue;
            }
        }
        uint32_t run_u = 0, run_u2 = 0, run_f = 0;
        int wslot = 0;  // ring slot of the next row to stage
        struct Seen { int flags, lo, hi; };
        // (stagers only) one batch: classify, convert, prefix, write into the ring, publish what was seen
        auto stage_batch = [&](int n0, const float (&v)[B], int parity) {
            // classification on the float samples: min / max of x (trunc is monotonic, so the smallest and largest trunc(x) are
            // those of the smallest and largest x; v_min / v_max skip a NaN: the magnitude test is written so that it catches it)
            bool frac = false, absurd = false;
            float xlo = INFINITY, xhi = -INFINITY;
            uint32_t q[B], q2[B], qf[BOTH ? B : 1];
            auto rows = [&](auto inside_tag) {
                constexpr bool INSIDE = decltype(inside_tag)::value;  // every row of the batch is inside the block's view
#pragma unroll
                for (int r = 0; r < B; ++r) {
                    const int gy = gy0 + n0 + r;
                    const bool ok = INSIDE ? col_ok : (col_ok && gy >= rmin && gy < rmax);
                    float x;
                    if (INSIDE) {
                        x = ok ? v[r] : 0.0f;
                    } else {
                        const bool pad = !col_ok || gy < 0 || gy >= p.gny;  // outside the DEM: a sample of elevation 0
                        x = ok ? v[r] : (pad ? 0.0f : (float)ci);
                    }
                    const int t = (int)x;
                    frac |= x != (float)t;
                    absurd |= !(fabsf(x) < kAbsLim + 1.0f);  // (NaN too)
                    xlo = fminf(xlo, x);
                    xhi = fmaxf(xhi, x);
                    const uint32_t u = (uint32_t)(t - ci);
                    run_u += u;
                    run_u2 += (uint32_t)__mul24((int)u, (int)u);
                    q[r] = run_u;
                    q2[r] = run_u2;
                    if (BOTH) {
                        run_f += stage_value<kStF>(ok ? x : 0.0f, 0.0f, 0);
                        qf[r] = run_f;
                    }
                }
            };
            if (gy0 + n0 >= rmin && gy0 + n0 + B <= rmax) rows(std::true_type{});
            else rows(std::false_type{});
#pragma unroll
            for (int r = 0; r < B; ++r) {
                int sl = wslot + r;
                sl = sl >= R ? sl - R : sl;
                Q[sl * PITCH + sdw] = q[r];
                Q[sl * PITCH + G::W + sdw] = q2[r];
                if (BOTH) Q[sl * PITCH + 2 * G::W + sdw] = qf[r];
            }
            // (a lane that saw NaN only keeps +-inf here: its flag is up; the clamps keep the range arithmetic in int32)
            int lo = (int)fminf(fmaxf(xlo, -(float)kBig), (float)kBig), hi = (int)fminf(fmaxf(xhi, -(float)kBig), (float)kBig);
            lo = wave_min_max<false>(lo);
            hi = wave_min_max<true>(hi);
            int fl = 0;
            if (__builtin_amdgcn_ballot_w64(frac)) fl |= kTileFrac;
            if (__builtin_amdgcn_ballot_w64(absurd)) fl |= kTileFloat;
            if (lane == 0) {
                int* w = wflags + (parity * NW + wave) * 4;
                w[0] = fl;
                w[1] = lo;
                w[2] = hi;
            }
        };
        auto advance_wslot = [&]() {
            wslot += B;
            wslot = wslot >= R ? wslot - R : wslot;
        };
        unsigned hist = 0;
        int hlo[HIST], hhi[HIST];
#pragma unroll
        for (int k = 0; k < HIST; ++k) hlo[k] = kBig, hhi[k] = -kBig;
        auto fold = [&](int parity, int nbatches) {
            // (the prologue publishes the range of all its batches at once: nbatches entries of the history)
            int fl = 0, lo = kBig, hi = -kBig;
#pragma unroll
            for (int w = 0; w < C::SW; ++w) {
                const int* q = wflags + (parity * NW + w) * 4;
                fl |= q[0];
                lo = min(lo, q[1]);
                hi = max(hi, q[2]);
            }
            fl = __builtin_amdgcn_readfirstlane(fl);
            constexpr int kStops = BOTH ? kTileFloat : (kTileFrac | kTileFloat);
            hist = ((hist << 1) | ((fl & kStops) ? 1u : 0u)) & ((1u << HIST) - 1u);
#pragma unroll
            for (int k = HIST - 1; k > 0; --k) hlo[k] = hlo[k - 1], hhi[k] = hhi[k - 1];
            hlo[0] = __builtin_amdgcn_readfirstlane(lo);
            hhi[0] = __builtin_amdgcn_readfirstlane(hi);
            (void)nbatches;
        };

        // prologue: the window of phase 0, one batch at a time; the stagers fold what they saw into one entry
        // (two batches of loads in flight - a second buffer, the phases unrolled in pairs - changed nothing: 2.27 -> 2.35 ms at
        // 7 px; the kernel does not wait for its loads)
        float va[B];
        __syncthreads();  // (the previous run's chain waves are done with the ring and the flag words)
        if (stager) {
            int fl = 0, lo = kBig, hi = -kBig;
#pragma unroll 1
            for (int k = 0; k < C::NB_PRO; ++k) {
                load_batch(k * B, va);
                stage_batch(k * B, va, 1);
                advance_wslot();
                if (lane == 0) {  // (own words: read back what stage_batch just published)
                    const int* w = wflags + (1 * NW + wave) * 4;
                    fl |= w[0];
                    lo = min(lo, w[1]);
                    hi = max(hi, w[2]);
                }
            }
            if (lane == 0) {
                int* w = wflags + (0 * NW + wave) * 4;
                w[0] = fl;
                w[1] = lo;
                w[2] = hi;
            }
            load_batch(C::PRO, va);
        } else {
            for (int k = 0; k < C::NB_PRO; ++k) advance_wslot();
        }
        __syncthreads();
        fold(0, C::NB_PRO);

        const int ocol = gx0 + lane * NCR;
        const bool lane_ok = lane >= DL && lane < DL + G::NVL;
        int tmode = kTileDone;
#ifdef STD_STAMPS
        long long ssum[6] = {0, 0, 0, 0, 0, 0};
#define SPEC_STAMP(i) { const long long now_ = __builtin_amdgcn_s_memtime(); ssum[i] += now_ - slast; slast = now_; }
        long long slast = __builtin_amdgcn_s_memtime();
#else
#define SPEC_STAMP(i)
#endif
        auto phase = [&](const int ph, float (&vbuf)[B]) {
            const int tile = tile0 + ph / PPT;
            int wlo = kBig, whi = -kBig;
#pragma unroll
            for (int k = 0; k < HIST; ++k) wlo = min(wlo, hlo[k]), whi = max(whi, hhi[k]);
            if (ph % PPT == 0) {
                // a new tile: the offset follows the range (std_ring_kernel; the padding's zeros belong to a border tile's range)
                const int oy0 = oyS + (ph / PPT) * C::TH;
                const bool at_border = cols_border || oy0 - G::M < 0 || oy0 + C::TH - 1 + G::M > p.gny - 1;
                const int tlo = at_border ? min(wlo, 0) : wlo, thi = at_border ? max(whi, 0) : whi;
                const bool too_wide = at_border && tlo <= thi && thi - tlo > 2 * lim32;
                const int mid = tlo + (thi - tlo) / 2;
                const bool fits_now = thi - ci <= lim32 && ci - tlo <= lim32;
                if (tlo <= thi && !too_wide && (abs(mid - ci) > lim32 / 4 || (!fits_now && thi - tlo <= 2 * lim32))) {
                    // (nobody writes the ring here: the stagers are in this branch too)
                    const uint32_t delta = (uint32_t)(mid - ci);
                    const uint32_t d2 = delta * delta;
                    for (int idx = threadIdx.x; idx < R * G::W; idx += NW * 64) {
                        const int sl = idx / G::W, col = idx - sl * G::W;
                        int since = sl - wslot;
                        since = since < 0 ? since + R : since;
                        const uint32_t k = (uint32_t)since + 1u;
                        uint32_t* q = Q + sl * PITCH + col;
                        const uint32_t q1 = q[0];
                        q[G::W] = q[G::W] - 2u * delta * q1 + d2 * k;
                        q[0] = q1 - delta * k;
                    }
                    if (stager) {
                        run_u2 = run_u2 - 2u * delta * run_u + d2 * (uint32_t)R;
                        run_u = run_u - delta * (uint32_t)R;
                    }
                    ci = mid;
                    __syncthreads();
                }
                if (BOTH) {
                    tmode = candidate(tile) && !too_wide ? kNeedsFraction : kTileDone;  // (kTileDone here: not this pass's; the mark stays)
                } else {
                    tmode = too_wide ? kTileGeneralWide : kTileDone;
                    if (threadIdx.x == 0) p.defer[tile] = (uint8_t)tmode;
                }
            }
            const bool fits = hist == 0 && whi - ci <= lim32 && ci - wlo <= lim32;
            if (BOTH) {
                if (tmode == kNeedsFraction && !fits) tmode = kTileGeneral;  // the mark stays: general kernel
            } else if ((tmode & 1) == 0 && !fits) {
                tmode = hist != 0 ? kTileGeneral : kTileGeneralWide;
                if (threadIdx.x == 0) p.defer[tile] = (uint8_t)tmode;
            }
            if (!BOTH && ph % PPT == PPT - 1 && tmode == kTileGeneral) ++seen_general;
            if (BOTH && ph % PPT == PPT - 1 && tmode == kNeedsFraction && threadIdx.x == 0) p.defer[tile] = kTileDone;
            if (stager) {
                // the batch phase ph + 1 needs, into the slots behind this phase's window; then the loads of the one after
                SPEC_STAMP(0)
                stage_batch(C::PRO + ph * B, vbuf, (ph + 1) & 1);
                SPEC_STAMP(1)
                load_batch(C::PRO + (ph + 1) * B, vbuf);
                SPEC_STAMP(2)
            } else if (BOTH ? tmode == kNeedsFraction : tmode == kTileDone) {
                SPEC_STAMP(0)
#pragma unroll 1
                for (int k = 0; k < C::RPW; ++k) {
                    const int j = (wave - C::SW) * C::RPW + k;  // row of the phase
                    int s0 = C::PAD - 1 + ph * B + j;           // stream row of Q index 0 of the output row's window
                    s0 = s0 % R;
                    uint32_t su[NCR], ctr[NCR], su2[NCR], dummy[NCR];
                    // (unrolling the two rows and dropping the scheduling fences, so that their chains interleave, changed nothing
                    // at 5 - 9 px and cost 4 - 9 % at 11 - 13 px)
                    ring_disc_sum<SIZE, NCR, R, 2, PITCH>(Q, s0, lane, su, ctr, false);
                    ring_disc_sum<SIZE, NCR, R, 2, PITCH>(Q + G::W, s0, lane, su2, dummy, false);
                    const int oy = oyS + ph * B + j;
                    if constexpr (BOTH) {
                        uint32_t sfi[4];
                        ring_disc_sum<SIZE, NCR, R, 2, PITCH>(Q + 2 * G::W, s0, lane, sfi, dummy, false);
                        if (lane_ok && oy >= p.out_row0 && oy < p.out_row0 + p.out_rows && ocol < p.nx) {
                            const size_t o = (size_t)(oy - p.out_row0) * p.nx + ocol;
                            Vec4<float> xs{{0.f, 0.f, 0.f, 0.f}};
                            if (WANT_TPI) xs = *reinterpret_cast<const Vec4<float>*>(p.in + (size_t)(oy - p.in_row0) * p.nx + ocol);
                            Vec4<float> out_s, out_t;
                            const double cd = (double)ci;
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                // the general kernel's expressions (disc_wave_kernel, finalise_row) on the same exact sums
                                const double dsu = (double)(int)su[t], dsu2 = (double)su2[t];
                                const double sf = (double)(int)sfi[t] * (1.0 / 65536.0);
                                const double s1 = (dsu + cd * n) + sf;
                                if (sfi[t] == 0) {
                                    out_s.v[t] = std_from_int_sums((int)su[t], (uint64_t)su2[t], (uint32_t)G::T.taps, (float)inv_nn1);
                                } else {
                                    const double s2 = dsu2 + 2.0 * cd * dsu + cd * cd * n;
                                    out_s.v[t] = std_from_sums(s1, s2, inv_n, inv_nm1);
                                }
                                if (WANT_TPI) out_t.v[t] = (float)((double)xs.v[t] - (s1 - (double)xs.v[t]) * inv_nm1);
                            }
                            *reinterpret_cast<Vec4<float>*>(p.sd + o) = out_s;
                            if (WANT_TPI) *reinterpret_cast<Vec4<float>*>(p.tpi + o) = out_t;
                        }
                    } else if (lane_ok && oy >= p.out_row0 && oy < p.out_row0 + p.out_rows) {
#pragma unroll
                        for (int P = 0; P < NCR / 4; ++P) {  // 16-byte pieces of the lane's columns (nx % 4 == 0: a piece is inside or outside)
                            if (ocol + 4 * P < p.nx) {
                                const size_t o = (size_t)(oy - p.out_row0) * p.nx + ocol + 4 * P;
                                Vec4<float> out_s, out_t;
#pragma unroll
                                for (int t = 0; t < 4; ++t) {
                                    const int c = 4 * P + t;
                                    out_s.v[t] = std_from_int_sums((int)su[c], (uint64_t)su2[c], (uint32_t)G::T.taps, (float)inv_nn1);
                                    if (WANT_TPI) {
                                        const int xi = (int)ctr[c] + ci;
                                        out_t.v[t] = (float)((double)xi - (double)((int)su[c] + ci * G::T.taps - xi) * inv_nm1);
                                    }
                                }
                                *reinterpret_cast<Vec4<float>*>(p.sd + o) = out_s;
                                if (WANT_TPI) *reinterpret_cast<Vec4<float>*>(p.tpi + o) = out_t;
                            }
                        }
                    }
                }
            }
            SPEC_STAMP(3)
            advance_wslot();
            __syncthreads();
            SPEC_STAMP(4)
#ifndef SPEC_LAB_NOFOLD  // (lab: what the per-phase bookkeeping costs - tools/ubench/tpi_lab.hip -DSPEC_LAB_NOFOLD)
            fold((ph + 1) & 1, 1);
#endif
            SPEC_STAMP(5)
        };
#pragma unroll 1
        for (int ph = 0; ph < nphase; ++ph) phase(ph, va);
#ifdef STD_STAMPS
        if (lane == 0 && (blockIdx.x == 3 || blockIdx.x == 131) && (wave == 0 || wave == 5 || wave == 11) && pos == first)
            printf("blk %3d wave %2d phases %d: top of phase %lld | stage / - %lld | loads / - %lld | (chains) %lld | barrier %lld | fold %lld  (memtime ticks)\n",
                   (int)blockIdx.x, wave, nphase, ssum[0], ssum[1], ssum[2], ssum[3], ssum[4], ssum[5]);
#endif
        pos += run_tiles;
    }
    if (!BOTH && p.report != nullptr && vb == nb / 2 && threadIdx.x == 0) {
        __hip_atomic_store(p.report + 1, (uint32_t)seen_general, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(p.report, (uint32_t)(last > first ? last - first : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int SIZE, bool WANT_TPI, bool BOTH = false, int NCR = 4>
__global__ __launch_bounds__((StdSpecCfg<SIZE, BOTH, NCR>::NW * 64), (StdSpecCfg<SIZE, BOTH, NCR>::kWavesPerSimd)) void std_ring_spec_kernel(WaveArgs p, int tiles_x, int tiles_y, PartRun deal) {
    TOPO_RUN_ONE((std_ring_spec_body<SIZE, WANT_TPI, BOTH, NCR>));
}
template <int SIZE, bool WANT_TPI, bool BOTH = false, int NCR = 4>
__global__ __launch_bounds__((StdSpecCfg<SIZE, BOTH, NCR>::NW * 64), (StdSpecCfg<SIZE, BOTH, NCR>::kWavesPerSimd)) void std_ring_spec_kernel_parts(WaveParts ps, int tiles_x) {
    TOPO_RUN_PARTS((std_ring_spec_body<SIZE, WANT_TPI, BOTH, NCR>));
}

template <int SIZE, bool WANT_TPI, bool BOTH = false, int NCR = 4>
int launch_std_ring_spec(const Block& b, float* tpi_out, float* std_out) {
    using G = RGeo<SIZE, NCR>;
    using C = StdSpecCfg<SIZE, BOTH, NCR>;
    Context& c = ctx();
    WaveArgs a{b.in, tpi_out, std_out, b.in_rows, b.in_row0, b.gny, b.nx, b.out_row0, b.out_rows,
               nullptr, nullptr, nullptr, 0, 0, 0};
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        TOPO_HIP(hipFuncSetAttribute((const void*)std_ring_spec_kernel<SIZE, WANT_TPI, BOTH, NCR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        int nblk = 0;
        TOPO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)std_ring_spec_kernel<SIZE, WANT_TPI, BOTH, NCR>, C::NW * 64, C::LDS));
        blocks_per_cu = nblk < 1 ? 1 : (nblk > 2 ? 2 : nblk);
    }
    WaveParts ps;
    int tiles_x = 0;
    long ntiles = 0;
    a.report = BOTH ? nullptr : dem_memo_report(b);
    TOPO_TRY(make_parts(b, a, C::TH, G::TILE_W, true, false, &ps, &tiles_x, &ntiles));
    const long grid = march_grid(c, blocks_per_cu, ntiles);
    deal_parts(&ps, tiles_x, grid, blocks_per_cu);
    return launch_parts(std_ring_spec_kernel<SIZE, WANT_TPI, BOTH, NCR>, std_ring_spec_kernel_parts<SIZE, WANT_TPI, BOTH, NCR>, grid, C::NW * 64, C::LDS, ps, tiles_x);
}

template <int SIZE, bool WANT_TPI, int MODE = kStdMain>
int launch_std_ring(const Block& b, float* tpi_out, float* std_out) {
    using G = RGeo<SIZE, 4>;
    using C = StdRingCfg<SIZE>;
    constexpr size_t kLds = C::LDS + (MODE == kStdBoth ? (size_t)C::R * G::W * sizeof(uint32_t) : 0);
    static_assert(kLds <= 160 * 1024, "the three images do not fit LDS");
    Context& c = ctx();
    WaveArgs a{b.in, tpi_out, std_out, b.in_rows, b.in_row0, b.gny, b.nx, b.out_row0, b.out_rows,
               nullptr, nullptr, nullptr, 0, 0, 0};
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        TOPO_HIP(hipFuncSetAttribute((const void*)std_ring_kernel<SIZE, WANT_TPI, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)kLds));
        int nblk = 0;
        TOPO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)std_ring_kernel<SIZE, WANT_TPI, MODE>, C::NW * 64,
                                                              kLds));
        blocks_per_cu = nblk < 1 ? 1 : (nblk > 2 ? 2 : nblk);  // small discs: two rings per CU
    }
    WaveParts ps;
    int tiles_x = 0;
    long ntiles = 0;
    a.report = MODE == kStdMain ? dem_memo_report(b) : nullptr;
    TOPO_TRY(make_parts(b, a, C::TH, G::TILE_W, true, false, &ps, &tiles_x, &ntiles));
    const long grid = march_grid(c, blocks_per_cu, ntiles);
    deal_parts(&ps, tiles_x, grid, blocks_per_cu);
    TOPO_TRY(launch_parts(std_ring_kernel<SIZE, WANT_TPI, MODE>, std_ring_kernel_parts<SIZE, WANT_TPI, MODE>, grid, C::NW * 64, kLds, ps, tiles_x));
    return TOPO_AMD_OK;
}

}  // namespace

}  // namespace topo
