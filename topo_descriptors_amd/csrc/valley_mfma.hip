// K7m: valley / ridge index on the matrix pipe, for rotated kernels of up to kValleyStreamMaxKernel cells a side (kernels of up to
// ~85 px) when the tables are point-symmetric, of up to kValleyMfmaMaxKernel otherwise.
//
// The same quantity as valley.hip (reference topo.py:431-447; the host's tables are the plane sums of the rotated
// kernels, flipped: a correlation): per pixel, the largest response over the angles of the maximum over the planes,
// and the angle it came from.  valley.hip evaluates the non-zero taps one v_fma_f32 each and runs at the rate at which
// the chip issues that instruction (39.9 T FMA/s: 523 ms for 180 angles x 3 planes of the 7 px kernel on 32768^2).
// Here the window of a pixel is the K dimension of a product
//     response[filter, pixel] = sum_k taps[filter, k] * z[pixel, k]
// with M = 32 filters ((angle, plane) pairs), N = 32 pixels (32 neighbours in a row) and K = 16 cells per
// v_mfma_f32_32x32x16_f16.  K runs over the LIVE cells only: the cells of the largest rotated canvas in which some
// filter has a non-zero tap - 53 of the 10 x 10 cells of the 7 px kernel (the reference masks the ring its spline
// rotation contaminates), 4 K steps.  A filter's own zeros inside the live set are multiplied: 64 slots against 36
// non-zero taps, on a pipe that is 32 x as fast as the vector ALU.
//
// float32 accuracy from f16 operands: z = hi + lo and tap = hi + lo (two f16 each, 22 bits), three products
// hi*hi + hi*lo + lo*hi accumulated in float32; the dropped lo*lo is 2^-22 of a term (measured: closer to the float64
// evaluation than valley.hip's float32 chain).  z is the standardised DEM (|z| of a few units), so no reference value is
// subtracted and a result does not depend on the tile a pixel falls in: row blocks keep the bits of the single block.
//
// Operands.  A (taps): laid out by the host as the instruction wants it, 1 KB per (filter tile, K step, part), streamed
// through LDS in groups of filter tiles (double-buffered global_load ... lds, one barrier per group; the four waves of a
// block share every fragment).  B (pixels): built ONCE per pixel tile from the f16 image of the block's DEM tile in LDS
// (ds_read_u16 at the live cells' offsets, which the kernel keeps in LDS: lanes 32-63 supply the second eight cells of a
// K step) and kept in registers for all the filters: 2 * KS * 4 registers per pixel tile, four pixel tiles (two, one: by KS) a wave
// at a time - every tap fragment read from LDS then feeds that many MFMAs.  The slots behind the last
// live cell re-read the first one against zero taps, so a non-finite sample reaches exactly the pixels whose live
// window holds it.  Two blocks of four waves share a CU.
// Measured (profiles/r06_valley_mfma.txt): the chip clocks at 1.9 GHz under this kernel and the matrix pipe is busy 75 - 78 % of
// the time - the kernel's time is the SUM of its MFMA cycles (32 each) and its vector instructions' issue cycles (4 each): the two
// waves of a SIMD do not overlap one's MFMAs with the other's vector instructions, whatever their priorities.  Taking a tile's
// maxima under the next tile's MFMAs (two result sets), reading a tile's fragments in one batch, and hand-placed scheduling
// groups each changed nothing; fewer K steps and more MFMAs per fragment read did.
//
// Result layout of the instruction: a lane holds, for pixel column (lane & 31), the 16 filters 8 q + 4 (lane >> 5) + i.
// The host puts the planes of an angle side by side in those 16 and the angles in rising order down the tiles: the
// maximum over the planes is one v_max3_f32 in a lane, the running best with its angle two registers per pixel tile,
// and "the first angle that reaches the maximum keeps it" (topo.py:438) is the strict comparison plus one exchange
// between the two lane halves at the end.
//
// Pixels whose live window holds a non-finite sample (or one beyond the f16 range after standardising: |z| > 65504)
// come out non-finite for every filter - 0 * inf - and are handed to valley.hip's kernel: this kernel stores norm = -1
// (the norm proper is clipped at 0) and raises its tile's flag; the launcher then runs the direct kernel over the
// flagged tiles, which rewrites exactly the marked pixels.  Which pixels those are depends on their own window alone.
//
// Three kernels: valley_mfma_kernel over the live cells (any tables), valley_fold_kernel over PAIRS of opposite cells, which
// tables that are point-symmetric bit by bit - the reference's - take: half the K steps (further down, "the folded form"), and
// valley_fold_stream_kernel, the folded form for kernels of 19 px and more, whose pixel operands no longer fit the registers.
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace topo {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWaves = 4;  // one per SIMD
constexpr int kThreads = 64 * kWaves;
constexpr int kTW = 64;                   // pixel columns of a block: two pixel tiles of 32
constexpr int kTH = kValleyMfmaTileRows;  // pixel rows of a block: eight per wave
constexpr int kPitch = kTW + kValleyMfmaMaxKernel - 1;  // halfs a row of the LDS image (even)
constexpr int kFragBytes = 1024;          // one operand fragment: 64 lanes x 8 halfs
constexpr int kMaxSteps = 15;             // K steps the kernel is built for (240 live cells)
static_assert(kPitch % 2 == 0, "the LDS image's rows start on a dword");

__host__ __device__ constexpr int pixel_tiles(int ks) { return ks <= 4 ? 4 : ks <= 8 ? 2 : 1; }  // a wave holds at a time
__host__ __device__ constexpr int group_tiles(int ks) {  // per LDS stage: at most 30 KB; even, or one
    return ks <= 3 ? 4 : ks <= 7 ? 2 : 1;
}
__host__ __device__ constexpr int image_bytes(int w) {
    return ((kTH + w - 1) * kPitch * 2 * 2 + kMaxSteps * 16 * 4 + kFragBytes - 1) / kFragBytes * kFragBytes;
}

struct VmArgs {
    const float* in;
    float* norm;
    float* dir;
    const unsigned char* atab;  // [group][tile in group][K step][hi, lo][64 lanes x 16 bytes]
    const int* koff;            // [K step][16]: the LDS distance (halfs) of the step's cells from the window's first cell
    const float* angles;
    int* flags;                 // one per block: some pixel left to the direct kernel
    int n_angles, n_groups;
    int w;                      // side of the window = of the largest rotated canvas
    int in_rows, in_row0, gny, nx;
    int out_row0, out_rows;
    float mean, stdev;
};

// the maximum over the planes of an angle.  Plain fmaxf, which returns the other operand for a NaN like valley.hip's: no
// inline assembly here - the compiler's hazard recogniser does not look into it, and a vector instruction that reads an
// MFMA result needs up to 11 wait states behind the MFMA (with v_max3_f32 in an asm statement the single-accumulator
// form of this kernel read results that were not there yet).
__device__ __forceinline__ float max_planes(const f32x16& acc, int first, int np) {
    float m = acc[first];
    for (int q = 1; q < np; ++q) m = __builtin_fmaxf(m, acc[first + q]);
    return m;
}

// The running best of a pixel tile in a lane: the largest response so far, the tile (its first position in the lane) it came
// from and - instead of the angle - that tile's maxima over the planes, one per angle: a tile then costs its APH plane maxima, the
// maximum of those, ONE comparison and APH + 1 selects instead of a comparison and a select per angle (16 vector instructions
// against 21 at three planes); which angle it was is read off once, at the end (`position`: the first that equals the best, so
// "the first angle that reaches the maximum keeps it" holds inside a tile as it does across tiles by the strict comparison; a NaN
// compares false and leaves the best alone, like valley.hip's).  For one plane a lane holds 16 angles: too many registers, the
// angle is tracked directly.
template <int NP>
struct Best {
    static constexpr int APH = 16 / NP;
    static constexpr bool kByTile = APH <= 8;
    float value;
    int first;                     // position of the winning tile's first angle (kByTile), or of the winning angle
    float of_tile[kByTile ? APH : 1];
    __device__ __forceinline__ void reset(int position0) {
        value = -INFINITY;
        first = position0;
#pragma unroll
        for (int s = 0; s < (kByTile ? APH : 1); ++s) of_tile[s] = -INFINITY;
    }
    __device__ __forceinline__ void take(const f32x16& acc, int tile_first) {
        if (kByTile) {
            float m[APH];
#pragma unroll
            for (int s = 0; s < APH; ++s) m[s] = max_planes(acc, s * NP, NP);
            float t = m[0];
#pragma unroll
            for (int s = 1; s < APH; ++s) t = __builtin_fmaxf(t, m[s]);
            const bool better = t > value;  // strict: an earlier tile keeps a tie
            value = __builtin_fmaxf(value, t);
            first = better ? tile_first : first;
#pragma unroll
            for (int s = 0; s < APH; ++s) of_tile[s] = better ? m[s] : of_tile[s];
        } else {
#pragma unroll
            for (int s = 0; s < APH; ++s) {
                const float m = max_planes(acc, s * NP, NP);
                const bool better = m > value;
                value = __builtin_fmaxf(value, m);
                first = better ? tile_first + s : first;
            }
        }
    }
    __device__ __forceinline__ int position() const {
        if (!kByTile) return first;
        int slot = 0;
#pragma unroll
        for (int s = APH - 1; s >= 0; --s) slot = of_tile[s] == value ? s : slot;
        return first + slot;
    }
};

template <int KS, int NP>
__global__ __launch_bounds__(kThreads, 2) void valley_mfma_kernel(VmArgs p) {
    constexpr int P = pixel_tiles(KS);
    constexpr int GT = group_tiles(KS);
    constexpr int GROUP_BYTES = GT * KS * 2 * kFragBytes;
    constexpr int UNITS = (kTH / kWaves) * 2 / P;  // a wave's 16 pixel tiles (8 rows x 2 halves of the 64 columns), P at a time
    constexpr int APH = 16 / NP;  // angles in a lane's 16 results
    constexpr int APT = 2 * APH;  // angles of a filter tile
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int rows_l = kTH + p.w - 1, cols_l = kTW + p.w - 1;
    const int plane = rows_l * kPitch;  // halfs: the hi plane, then the lo plane, then the cells' offsets
    _Float16* img = reinterpret_cast<_Float16*>(lds);
    int* koff = reinterpret_cast<int*>(lds + (size_t)plane * 4);
    unsigned char* abuf = lds + image_bytes(p.w);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = lane & 31;
    const int h = lane >> 5;
    const int ox0 = blockIdx.x * kTW;
    const int oy0 = p.out_row0 + blockIdx.y * kTH;
    const int reach = p.w / 2;  // "same" centring: a kernel of side K starts K / 2 before the pixel

    // the stream of tap fragments: group g of the table into buffer `buf`, fragment by fragment over the waves
    auto issue_group = [&](int g, int buf) {
        const unsigned char* src = p.atab + (size_t)g * GROUP_BYTES + lane * 16;
        unsigned char* dst = abuf + buf * GROUP_BYTES;
#pragma unroll
        for (int f = wave; f < GT * KS * 2; f += kWaves)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * kFragBytes),
                                             (__attribute__((address_space(3))) void*)(dst + f * kFragBytes + lane * 16), 16, 0,
                                             0);
    };
    issue_group(0, 0);

    // the block's DEM tile, standardised the way numpy does it ((x - mean) / std in float32), zero outside the DEM
    // (the reference zero-pads the standardised field), as two f16 planes
    for (int r = wave; r < rows_l; r += kWaves) {
        const int gy = oy0 - reach + r;
        const int by = gy - p.in_row0;
        const bool row_ok = gy >= 0 && gy < p.gny && by >= 0 && by < p.in_rows;
        for (int c = lane; c < cols_l; c += 64) {
            const int gx = ox0 - reach + c;
            float v = 0.0f;
            if (row_ok && gx >= 0 && gx < p.nx) v = (p.in[(size_t)by * p.nx + gx] - p.mean) / p.stdev;
            const _Float16 hi = (_Float16)v;
            img[r * kPitch + c] = hi;
            img[plane + r * kPitch + c] = (_Float16)(v - (float)hi);
        }
    }
    for (int i = threadIdx.x; i < KS * 16; i += kThreads) koff[i] = p.koff[i];

    int it = 0;  // position in the stream: buffer it & 1 holds group it % n_groups
#pragma unroll 1
    for (int u = 0; u < UNITS; ++u) {
        f16x8 bh[P][KS], bl[P][KS];
        Best<NP> best[P];
#pragma unroll
        for (int pt = 0; pt < P; ++pt) best[pt].reset(0);
        if (u == 0) __syncthreads();  // the image tile and the offsets are written
        {
            const int* ko = koff + 8 * h;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int off = ko[s * 16 + e];
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) {
                        const int q = u * P + pt;  // the wave's pixel tile: row wave + 4 (q / 2), columns 32 (q % 2) ...
                        const _Float16* at = img + (wave + kWaves * (q >> 1)) * kPitch + 32 * (q & 1) + n + off;
                        bh[pt][s][e] = at[0];
                        bl[pt][s][e] = at[plane];
                    }
                }
        }
#pragma unroll 1
        for (int g = 0; g < p.n_groups; ++g, ++it) {
            // my fragments of this group have landed; behind the barrier everybody's have, and nobody reads the other buffer any more
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            issue_group(g + 1 == p.n_groups ? 0 : g + 1, (it + 1) & 1);
            const unsigned char* ab = abuf + (it & 1) * GROUP_BYTES + lane * 16;
#pragma unroll
            for (int tt = 0; tt < GT; ++tt) {
                const int tile = g * GT + tt;  // (the host fills the last group up with copies of the last angle)
                f32x16 acc[P];
#pragma unroll
                for (int pt = 0; pt < P; ++pt)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[pt][v] = 0.0f;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(ab + ((tt * KS + s) * 2) * kFragBytes);
                    const f16x8 al = *reinterpret_cast<const f16x8*>(ab + ((tt * KS + s) * 2 + 1) * kFragBytes);
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[pt][s], acc[pt], 0, 0, 0);
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[pt][s], acc[pt], 0, 0, 0);
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[pt][s], acc[pt], 0, 0, 0);
                }
                // the maxima over the planes of the APH angles a lane holds against the running best (struct Best)
#pragma unroll
                for (int pt = 0; pt < P; ++pt) best[pt].take(acc[pt], tile * APT + h * APH);
            }
        }
        // lanes l and l + 32 hold different angles of the same pixel: the larger one, the earlier angle on a tie
        bool unfinished = false;
#pragma unroll
        for (int pt = 0; pt < P; ++pt) {
            const float mine = best[pt].value;
            const int mi = best[pt].position();
            const float ob = __shfl_xor(mine, 32);
            const int oi = __shfl_xor(mi, 32);
            const bool take = ob > mine || (ob == mine && oi < mi);
            const float b = take ? ob : mine;
            const int bi = min(take ? oi : mi, p.n_angles - 1);  // (the last tile is filled up with copies of the last angle)
            const int q = u * P + pt;
            const int ox = ox0 + 32 * (q & 1) + n;
            const int oy = oy0 + wave + kWaves * (q >> 1);
            if (ox >= p.nx || oy >= p.out_row0 + p.out_rows) continue;
            const bool finite = fabsf(b) < INFINITY;
            unfinished = unfinished || !finite;
            const size_t o = (size_t)(oy - p.out_row0) * p.nx + ox;
            if (h == 0)
                p.norm[o] = finite ? fmaxf(b, 0.0f) : -1.0f;  // clip(min=0), topo.py:446; -1: left to the direct kernel
            else
                p.dir[o] = finite ? p.angles[bi] : 0.0f;
        }
        if (unfinished) p.flags[blockIdx.y * gridDim.x + blockIdx.x] = 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stream's last, unused group: not into the LDS of the next block
}

// ---- the folded form ------------------------------------------------------------------------------------------------
// The reference's kernels are V / U profiles across one axis, constant along the other, turned about the canvas centre: exactly
// point-symmetric, K(c) = K(c*) with c* the cell opposite the centre (the launcher checks the tables bit by bit; any other table
// takes the kernel above).  Then sum_c K(c) z(c) = sum over PAIRS K(c) (z(c) + z(c*)): K runs over the live pairs - 27 for the
// 7 px kernel, 2 K steps instead of 4.  Canvases of odd and of even side have different centres (half a cell apart in the common
// window), so there are two classes of angles, each with its own pairs: filter tiles are pure in class, a wave builds its pixel
// operands once per class (z(c) + z(c*) in float32 from a float32 image of the DEM tile, then the hi / lo split), and runs the
// class's tiles.  Tiles are then no longer in angle order: the running best is kept per class (rising angles inside a class, the
// strict comparison as before) and the classes - like the two lane halves - are merged on (value, then the smaller angle index).
struct VfArgs {
    const float* in;
    float* norm;
    float* dir;
    const unsigned char* atab;  // [group][tile in group][K step][hi, lo][64 lanes x 16 bytes]; the groups of class 0, then of class 1
    const int* koff;            // [class][K step][16][2]: the LDS distances (floats) of a pair's two cells from the window's first cell
    const int* pos_angle;       // [group][tile][row position] -> index of the angle (copies of a class's last angle fill its last group)
    const float* angles;
    int* flags;
    int class_groups[2];
    int class_tiles[2];         // filter tiles of the class (its last group may hold fewer than the stream's stage does)
    int w;
    int in_rows, in_row0, gny, nx;
    int out_row0, out_rows;
    float mean, stdev;
};

__host__ __device__ constexpr int fold_image_bytes(int w) {
    return ((kTH + w - 1) * kPitch * 4 + 2 * kMaxSteps * 16 * 2 * 4 + kFragBytes - 1) / kFragBytes * kFragBytes;
}

template <int KS, int NP>
__global__ __launch_bounds__(kThreads, 2) void valley_fold_kernel(VfArgs p) {
    constexpr int P = pixel_tiles(KS);
    constexpr int GT = group_tiles(KS);
    constexpr int GROUP_BYTES = GT * KS * 2 * kFragBytes;
    constexpr int UNITS = (kTH / kWaves) * 2 / P;
    constexpr int APH = 16 / NP;
    constexpr int APT = 2 * APH;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int rows_l = kTH + p.w - 1, cols_l = kTW + p.w - 1;
    float* img = reinterpret_cast<float*>(lds);
    int* koff = reinterpret_cast<int*>(lds + (size_t)rows_l * kPitch * 4);
    unsigned char* abuf = lds + fold_image_bytes(p.w);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = lane & 31;
    const int h = lane >> 5;
    const int ox0 = blockIdx.x * kTW;
    const int oy0 = p.out_row0 + blockIdx.y * kTH;
    const int reach = p.w / 2;
    const int n_groups = p.class_groups[0] + p.class_groups[1];

    auto issue_group = [&](int g, int buf) {
        const unsigned char* src = p.atab + (size_t)g * GROUP_BYTES + lane * 16;
        unsigned char* dst = abuf + buf * GROUP_BYTES;
#pragma unroll
        for (int f = wave; f < GT * KS * 2; f += kWaves)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * kFragBytes),
                                             (__attribute__((address_space(3))) void*)(dst + f * kFragBytes + lane * 16), 16, 0,
                                             0);
    };
    issue_group(0, 0);

    for (int r = wave; r < rows_l; r += kWaves) {
        const int gy = oy0 - reach + r;
        const int by = gy - p.in_row0;
        const bool row_ok = gy >= 0 && gy < p.gny && by >= 0 && by < p.in_rows;
        for (int c = lane; c < cols_l; c += 64) {
            const int gx = ox0 - reach + c;
            float v = 0.0f;
            if (row_ok && gx >= 0 && gx < p.nx) v = (p.in[(size_t)by * p.nx + gx] - p.mean) / p.stdev;
            img[r * kPitch + c] = v;
        }
    }
    for (int i = threadIdx.x; i < 2 * KS * 16 * 2; i += kThreads) koff[i] = p.koff[i];

    int it = 0;
#pragma unroll 1
    for (int u = 0; u < UNITS; ++u) {
        float rv[P];  // the best over the classes done so far, and its angle's index
        int ri[P];
        bool bad[P];  // a class came out non-finite: a non-finite sample on one of ITS pairs (the other class may not see that cell)
#pragma unroll
        for (int pt = 0; pt < P; ++pt) {
            rv[pt] = -INFINITY;
            ri[pt] = 0x7fffffff;
            bad[pt] = false;
        }
        if (u == 0) __syncthreads();
        int g0 = 0;
#pragma unroll 1
        for (int cls = 0; cls < 2; ++cls) {
            const int ng = p.class_groups[cls];
            if (ng == 0) continue;  // (wave-uniform)
            f16x8 bh[P][KS], bl[P][KS];
            Best<NP> best[P];
#pragma unroll
            for (int pt = 0; pt < P; ++pt) best[pt].reset(g0 * GT * APT);
            {
                const int* ko = koff + (cls * KS * 16 + 8 * h) * 2;
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int o1 = ko[(s * 16 + e) * 2], o2 = ko[(s * 16 + e) * 2 + 1];
#pragma unroll
                        for (int pt = 0; pt < P; ++pt) {
                            const int q = u * P + pt;
                            const float* at = img + (wave + kWaves * (q >> 1)) * kPitch + 32 * (q & 1) + n;
                            const float zf = at[o1] + at[o2];
                            const _Float16 hi = (_Float16)zf;
                            bh[pt][s][e] = hi;
                            bl[pt][s][e] = (_Float16)(zf - (float)hi);
                        }
                    }
            }
#pragma unroll 1
            for (int g = g0; g < g0 + ng; ++g, ++it) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                issue_group(g + 1 == n_groups ? 0 : g + 1, (it + 1) & 1);
                const unsigned char* ab = abuf + (it & 1) * GROUP_BYTES + lane * 16;
#pragma unroll
                for (int tt = 0; tt < GT; ++tt) {
                    if ((g - g0) * GT + tt >= p.class_tiles[cls]) break;  // (wave-uniform: the class's last stage is not full)
                    f32x16 acc[P];
#pragma unroll
                    for (int pt = 0; pt < P; ++pt)
#pragma unroll
                        for (int v = 0; v < 16; ++v) acc[pt][v] = 0.0f;
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const f16x8 ah = *reinterpret_cast<const f16x8*>(ab + ((tt * KS + s) * 2) * kFragBytes);
                        const f16x8 al = *reinterpret_cast<const f16x8*>(ab + ((tt * KS + s) * 2 + 1) * kFragBytes);
#pragma unroll
                        for (int pt = 0; pt < P; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[pt][s], acc[pt], 0, 0, 0);
#pragma unroll
                        for (int pt = 0; pt < P; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[pt][s], acc[pt], 0, 0, 0);
#pragma unroll
                        for (int pt = 0; pt < P; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[pt][s], acc[pt], 0, 0, 0);
                    }
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) best[pt].take(acc[pt], (g * GT + tt) * APT + h * APH);
                }
            }
            g0 += ng;
#pragma unroll
            for (int pt = 0; pt < P; ++pt) {
                const float bv = best[pt].value;
                const int ai = p.pos_angle[best[pt].position()];
                bad[pt] = bad[pt] || !(fabsf(bv) < INFINITY);
                const bool take = bv > rv[pt] || (bv == rv[pt] && ai < ri[pt]);
                rv[pt] = take ? bv : rv[pt];
                ri[pt] = take ? ai : ri[pt];
            }
        }
        bool unfinished = false;
#pragma unroll
        for (int pt = 0; pt < P; ++pt) {
            const float ob = __shfl_xor(rv[pt], 32);
            const int oi = __shfl_xor(ri[pt], 32);
            const bool take = ob > rv[pt] || (ob == rv[pt] && oi < ri[pt]);
            const float b = take ? ob : rv[pt];
            const int bi = take ? oi : ri[pt];
            const int q = u * P + pt;
            const int ox = ox0 + 32 * (q & 1) + n;
            const int oy = oy0 + wave + kWaves * (q >> 1);
            if (ox >= p.nx || oy >= p.out_row0 + p.out_rows) continue;
            const bool finite = !bad[pt] && fabsf(b) < INFINITY;
            unfinished = unfinished || !finite;
            const size_t o = (size_t)(oy - p.out_row0) * p.nx + ox;
            if (h == 0)
                p.norm[o] = finite ? fmaxf(b, 0.0f) : -1.0f;
            else
                p.dir[o] = finite ? p.angles[bi] : 0.0f;
        }
        if (unfinished) p.flags[blockIdx.y * gridDim.x + blockIdx.x] = 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- the folded form for larger kernels: the pixel operands streamed too ----------------------------------------------
// Beyond 15 K steps (kernels of 19 px and more) a pixel tile's operands no longer fit the registers.  Here a wave works on
// kSTG filter tiles x kSP pixel tiles at a time (kSTG * kSP accumulators) and walks the K steps in chunks of kSKC: it builds the
// chunk's pixel operands (the fold and the split, as above), multiplies them into the kSTG tiles' accumulators, and moves on - the
// operands are rebuilt for every group of kSTG filter tiles.  The tap stream's stage is one (tile group, chunk).  One block of
// eight waves a CU (the float32 image of the DEM tile with the reach of a 120-cell canvas, the pairs' offsets and two stages are
// ~150 KB).  Per 32 pixels, filter tile and K step: 3 MFMAs (96 cycles) + a third of ~11 vector instructions' 44.  8192^2, 180
// angles x 3 planes: 3.6 - 3.9 ms per K step of pairs = 5 x the tap-by-tap kernel from 19 to 41 px (41 px: 1523 -> 291 ms), and
// faster than the FFT route's 2.05 s up to the 120-cell windows the LDS image holds (45 px 350 ms, 65 px 732, 85 px 1252;
// profiles/r06_valley_mfma.txt section 7).
constexpr int kSWaves = 8;
constexpr int kSThreads = 64 * kSWaves;
constexpr int kSP = 2;    // pixel tiles a wave holds at a time
constexpr int kSTG = 3;   // filter tiles it accumulates at a time
constexpr int kSKC = 2;   // K steps per chunk = per stage of the tap stream
constexpr int kStageBytes = kSTG * kSKC * 2 * kFragBytes;
constexpr int kStreamMaxSteps = 352;  // K steps (multiple of kSKC): 5632 live pairs a class (rotated kernels of up to ~120 cells a side)

struct VsArgs {
    const float* in;
    float* norm;
    float* dir;
    const unsigned char* atab;  // [class][tile group][chunk][tile in group][K step in chunk][hi, lo][64 lanes x 16 bytes]
    const unsigned short* koff; // [class][K step][16]: the LDS distance (floats) of a pair's first cell from the window's first cell ...
    int pair_sum[2];            // ... and of the two cells together (opposite cells: their distances add up to the same for a class)
    const int* pos_angle;       // [class][tile group][tile][row position] -> index of the angle
    const float* angles;
    int* flags;
    int class_tiles[2];
    int ks;                     // K steps (a multiple of kSKC)
    int w, pitch;               // window side; floats a row of the LDS image
    int in_rows, in_row0, gny, nx;
    int out_row0, out_rows;
    float mean, stdev;
};

__host__ __device__ inline int stream_image_bytes(int w, int pitch, int ks) {
    return ((kTH + w - 1) * pitch * 4 + 2 * ks * 16 * 2 + kFragBytes - 1) / kFragBytes * kFragBytes;
}

template <int NP>
__global__ __launch_bounds__(kSThreads, 1) void valley_fold_stream_kernel(VsArgs p) {
    constexpr int P = kSP;
    constexpr int UNITS = (kTH / kSWaves) * 2 / P;
    constexpr int APH = 16 / NP;
    constexpr int APT = 2 * APH;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int rows_l = kTH + p.w - 1, cols_l = kTW + p.w - 1;
    float* img = reinterpret_cast<float*>(lds);
    unsigned short* koff = reinterpret_cast<unsigned short*>(lds + (size_t)rows_l * p.pitch * 4);
    unsigned char* abuf = lds + stream_image_bytes(p.w, p.pitch, p.ks);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = lane & 31;
    const int h = lane >> 5;
    const int ox0 = blockIdx.x * kTW;
    const int oy0 = p.out_row0 + blockIdx.y * kTH;
    const int reach = p.w / 2;
    const int n_chunks = p.ks / kSKC;
    const int groups0 = (p.class_tiles[0] + kSTG - 1) / kSTG, groups1 = (p.class_tiles[1] + kSTG - 1) / kSTG;
    const int n_stages = (groups0 + groups1) * n_chunks;

    auto issue_stage = [&](int stage, int buf) {
        const unsigned char* src = p.atab + (size_t)stage * kStageBytes + lane * 16;
        unsigned char* dst = abuf + buf * kStageBytes;
#pragma unroll
        for (int f = wave; f < kSTG * kSKC * 2; f += kSWaves)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * kFragBytes),
                                             (__attribute__((address_space(3))) void*)(dst + f * kFragBytes + lane * 16), 16, 0,
                                             0);
    };
    issue_stage(0, 0);

    for (int r = wave; r < rows_l; r += kSWaves) {
        const int gy = oy0 - reach + r;
        const int by = gy - p.in_row0;
        const bool row_ok = gy >= 0 && gy < p.gny && by >= 0 && by < p.in_rows;
        for (int c = lane; c < cols_l; c += 64) {
            const int gx = ox0 - reach + c;
            float v = 0.0f;
            if (row_ok && gx >= 0 && gx < p.nx) v = (p.in[(size_t)by * p.nx + gx] - p.mean) / p.stdev;
            img[r * p.pitch + c] = v;
        }
    }
    for (int i = threadIdx.x; i < 2 * p.ks * 16; i += kSThreads) koff[i] = p.koff[i];

    int it = 0;  // position in the stream: buffer it & 1 holds stage it % n_stages
#pragma unroll 1
    for (int u = 0; u < UNITS; ++u) {
        float rv[P];
        int ri[P];
        bool bad[P];
        const float* at[P];
#pragma unroll
        for (int pt = 0; pt < P; ++pt) {
            rv[pt] = -INFINITY;
            ri[pt] = 0x7fffffff;
            bad[pt] = false;
            const int q = u * P + pt;  // the wave's pixel tile: row wave + 8 (q / 2), columns 32 (q % 2) ...
            at[pt] = img + (wave + kSWaves * (q >> 1)) * p.pitch + 32 * (q & 1) + n;
        }
        if (u == 0) __syncthreads();
        int stage = 0, tile0 = 0;
#pragma unroll 1
        for (int cls = 0; cls < 2; ++cls) {
            const int tiles = p.class_tiles[cls];
            Best<NP> best[P];
#pragma unroll
            for (int pt = 0; pt < P; ++pt) best[pt].reset(tile0 * APT);
#pragma unroll 1
            for (int tg = 0; tg * kSTG < tiles; ++tg) {
                f32x16 acc[kSTG][P];
#pragma unroll
                for (int t = 0; t < kSTG; ++t)
#pragma unroll
                    for (int pt = 0; pt < P; ++pt)
#pragma unroll
                        for (int v = 0; v < 16; ++v) acc[t][pt][v] = 0.0f;
#pragma unroll 1
                for (int c = 0; c < n_chunks; ++c, ++stage, ++it) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    issue_stage(stage + 1 == n_stages ? 0 : stage + 1, (it + 1) & 1);
                    // the chunk's pixel operands: z(c) + z(c*) in float32, split into two f16
                    f16x8 bh[P][kSKC], bl[P][kSKC];
                    const unsigned short* ko = koff + (cls * p.ks + c * kSKC) * 16 + 8 * h;
                    const int pair_sum = p.pair_sum[cls];
#pragma unroll
                    for (int s = 0; s < kSKC; ++s)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int o1 = ko[s * 16 + e], o2 = pair_sum - o1;
#pragma unroll
                            for (int pt = 0; pt < P; ++pt) {
                                const float zf = at[pt][o1] + at[pt][o2];
                                const _Float16 hi = (_Float16)zf;
                                bh[pt][s][e] = hi;
                                bl[pt][s][e] = (_Float16)(zf - (float)hi);
                            }
                        }
                    const unsigned char* ab = abuf + (it & 1) * kStageBytes + lane * 16;
#pragma unroll
                    for (int t = 0; t < kSTG; ++t) {
                        if (tg * kSTG + t >= tiles) break;  // (wave-uniform: the class's last group is not full)
#pragma unroll
                        for (int s = 0; s < kSKC; ++s) {
                            const f16x8 ah = *reinterpret_cast<const f16x8*>(ab + ((t * kSKC + s) * 2) * kFragBytes);
                            const f16x8 al = *reinterpret_cast<const f16x8*>(ab + ((t * kSKC + s) * 2 + 1) * kFragBytes);
#pragma unroll
                            for (int pt = 0; pt < P; ++pt)
                                acc[t][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[pt][s], acc[t][pt], 0, 0, 0);
#pragma unroll
                            for (int pt = 0; pt < P; ++pt)
                                acc[t][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[pt][s], acc[t][pt], 0, 0, 0);
#pragma unroll
                            for (int pt = 0; pt < P; ++pt)
                                acc[t][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[pt][s], acc[t][pt], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < kSTG; ++t) {
                    if (tg * kSTG + t >= tiles) break;
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) best[pt].take(acc[t][pt], (tile0 + tg * kSTG + t) * APT + h * APH);
                }
            }
            if (tiles > 0) {
#pragma unroll
                for (int pt = 0; pt < P; ++pt) {
                    const float bv = best[pt].value;
                    const int ai = p.pos_angle[best[pt].position()];
                    bad[pt] = bad[pt] || !(fabsf(bv) < INFINITY);
                    const bool take = bv > rv[pt] || (bv == rv[pt] && ai < ri[pt]);
                    rv[pt] = take ? bv : rv[pt];
                    ri[pt] = take ? ai : ri[pt];
                }
            }
            tile0 += (tiles + kSTG - 1) / kSTG * kSTG;
        }
        bool unfinished = false;
#pragma unroll
        for (int pt = 0; pt < P; ++pt) {
            const float ob = __shfl_xor(rv[pt], 32);
            const int oi = __shfl_xor(ri[pt], 32);
            const bool take = ob > rv[pt] || (ob == rv[pt] && oi < ri[pt]);
            const float b = take ? ob : rv[pt];
            const int bi = take ? oi : ri[pt];
            const int q = u * P + pt;
            const int ox = ox0 + 32 * (q & 1) + n;
            const int oy = oy0 + wave + kSWaves * (q >> 1);
            if (ox >= p.nx || oy >= p.out_row0 + p.out_rows) continue;
            const bool finite = !bad[pt] && fabsf(b) < INFINITY;
            unfinished = unfinished || !finite;
            const size_t o = (size_t)(oy - p.out_row0) * p.nx + ox;
            if (h == 0)
                p.norm[o] = finite ? fmaxf(b, 0.0f) : -1.0f;
            else
                p.dir[o] = finite ? p.angles[bi] : 0.0f;
        }
        if (unfinished) p.flags[blockIdx.y * gridDim.x + blockIdx.x] = 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// float -> f16 bits, round to nearest even (the host builds the tap operands; no _Float16 arithmetic in host code)
uint16_t f16_bits(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));
    if (x < 0x38800000u) {  // below 2^-14: a subnormal half, in units of 2^-24
        float a;
        std::memcpy(&a, &x, 4);
        return (uint16_t)(sign | (uint16_t)std::nearbyintf(a * 16777216.0f));  // 1024 = the smallest normal, as it should
    }
    const uint32_t r = x + 0xfffu + ((x >> 13) & 1u);
    if (r >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);
    return (uint16_t)(sign | ((r - 0x38000000u) >> 13));
}

float f16_value(uint16_t b) {
    const int e = (b >> 10) & 31, m = b & 1023;
    float v;
    if (e == 0)
        v = std::ldexp((float)m, -24);
    else if (e == 31)
        v = m ? NAN : INFINITY;
    else
        v = std::ldexp((float)(m | 1024), e - 25);
    return (b & 0x8000u) ? -v : v;
}

template <int KS, int NP>
int launch_ks(const VmArgs& a, dim3 grid) {
    const int lds = image_bytes(a.w) + 2 * group_tiles(KS) * KS * 2 * kFragBytes;
    TOPO_HIP(hipFuncSetAttribute((const void*)valley_mfma_kernel<KS, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((valley_mfma_kernel<KS, NP>), grid, dim3(kThreads), lds, ctx().compute, a);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

template <int KS>
int launch_np(const VmArgs& a, dim3 grid, int n_planes) {
    switch (n_planes) {
        case 1: return launch_ks<KS, 1>(a, grid);
        case 2: return launch_ks<KS, 2>(a, grid);
        case 3: return launch_ks<KS, 3>(a, grid);
        default: return launch_ks<KS, 4>(a, grid);
    }
}

template <int KS, int NP>
int launch_fold_ks(const VfArgs& a, dim3 grid) {
    const int lds = fold_image_bytes(a.w) + 2 * group_tiles(KS) * KS * 2 * kFragBytes;
    TOPO_HIP(hipFuncSetAttribute((const void*)valley_fold_kernel<KS, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((valley_fold_kernel<KS, NP>), grid, dim3(kThreads), lds, ctx().compute, a);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

template <int KS>
int launch_fold_np(const VfArgs& a, dim3 grid, int n_planes) {
    switch (n_planes) {
        case 1: return launch_fold_ks<KS, 1>(a, grid);
        case 2: return launch_fold_ks<KS, 2>(a, grid);
        case 3: return launch_fold_ks<KS, 3>(a, grid);
        default: return launch_fold_ks<KS, 4>(a, grid);
    }
}

template <int NP>
int launch_stream_np(const VsArgs& a, dim3 grid) {
    const int lds = stream_image_bytes(a.w, a.pitch, a.ks) + 2 * kStageBytes;
    TOPO_HIP(hipFuncSetAttribute((const void*)valley_fold_stream_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((valley_fold_stream_kernel<NP>), grid, dim3(kSThreads), lds, ctx().compute, a);
    TOPO_HIP(hipGetLastError());
    return TOPO_AMD_OK;
}

// The streamed folded form (launch_fold's classes and pairs): more than 15 K steps, or a window wider than the register-resident
// kernels' LDS image.
int launch_fold_stream(const Block& b, const float* taps, const int32_t* ksize, const float* angles, int n_angles, int n_planes, int W,
                       double mean, double stdev, float* norm_out, float* dir_out, const int** flags_out, int* flag_cols, int* folded,
                       int n_cls, const std::vector<int>& cls_of, const std::vector<int>& centre, const std::vector<int>* pair_of,
                       const std::vector<int>* pair_cell) {
    int KS = 0;
    for (int c = 0; c < n_cls; ++c) KS = std::max(KS, ((int)pair_cell[c].size() + 15) / 16);
    KS = (KS + kSKC - 1) / kSKC * kSKC;
    const int pitch = (kTW + W - 1 + 1) & ~1;
    if (KS > kStreamMaxSteps || W > kValleyStreamMaxKernel ||
        stream_image_bytes(W, pitch, KS) + 2 * kStageBytes > 160 * 1024)
        return TOPO_AMD_OK;
    const int aph = 16 / n_planes, apt = 2 * aph, n_chunks = KS / kSKC;
    std::vector<int> pos_angle;
    int class_tiles[2] = {0, 0}, class_groups[2] = {0, 0}, first_group[2] = {0, 0};
    for (int c = 0; c < n_cls; ++c) {
        std::vector<int> mine;
        for (int ang = 0; ang < n_angles; ++ang)
            if (cls_of[ang] == c) mine.push_back(ang);
        class_tiles[c] = ((int)mine.size() + apt - 1) / apt;
        class_groups[c] = (class_tiles[c] + kSTG - 1) / kSTG;
        first_group[c] = c == 0 ? 0 : class_groups[0];
        for (int k = 0; k < class_groups[c] * kSTG * apt; ++k) pos_angle.push_back(mine[std::min(k, (int)mine.size() - 1)]);
    }
    const int n_groups = class_groups[0] + class_groups[1];
    const size_t frag_halfs = kFragBytes / 2;
    std::vector<uint16_t> atab((size_t)n_groups * n_chunks * kSTG * kSKC * 2 * frag_halfs, 0);
    std::vector<size_t> first_tap((size_t)n_angles);
    {
        size_t at = 0;
        for (int ang = 0; ang < n_angles; ++ang) {
            first_tap[ang] = at;
            at += (size_t)ksize[ang] * ksize[ang] * 4;
        }
    }
    for (size_t pos = 0; pos < pos_angle.size(); ++pos) {
        const int ang = pos_angle[pos], c = cls_of[ang], C = centre[c];
        const int ks = ksize[ang], sh = W / 2 - ks / 2;
        const int tile = (int)pos / apt, hrow = ((int)pos % apt) / aph, slot = ((int)pos % apt) % aph;
        const int group = tile / kSTG, t_in = tile % kSTG;  // (groups run through both classes: a class's tiles fill whole groups)
        const float* t = taps + first_tap[ang];
        for (int ky = 0; ky < ks; ++ky)
            for (int kx = 0; kx < ks; ++kx) {
                const int wy = ky + sh, wx = kx + sh, cell = wy * W + wx, other = (C - wy) * W + (C - wx);
                if (cell > other) continue;
                const int k = pair_of[c][cell];
                if (k < 0) continue;
                for (int q = 0; q < n_planes; ++q) {
                    float v = t[((size_t)ky * ks + kx) * 4 + q];
                    if (v == 0.0f) continue;
                    if (cell == other) v *= 0.5f;
                    const uint16_t hi = f16_bits(v);
                    const uint16_t lo = f16_bits(v - f16_value(hi));
                    const int r = slot * n_planes + q, m = 8 * (r / 4) + 4 * hrow + r % 4;
                    const int s = k / 16, kh = (k % 16) / 8, e = k % 8;
                    const size_t stage = (size_t)group * n_chunks + s / kSKC;
                    const size_t frag = (stage * kSTG * kSKC + (size_t)t_in * kSKC + s % kSKC) * 2;
                    const size_t at = (size_t)(m + 32 * kh) * 8 + e;
                    atab[frag * frag_halfs + at] = hi;
                    atab[(frag + 1) * frag_halfs + at] = lo;
                }
            }
    }
    // one table of ints: the pairs' first cells (16 bits each, two to an int), then the positions' angles
    const size_t koff_ints = (size_t)2 * KS * 16 / 2;
    std::vector<int> ints(koff_ints, 0);
    {
        unsigned short* k16 = reinterpret_cast<unsigned short*>(ints.data());
        for (int c = 0; c < n_cls; ++c)
            for (int k = 0; k < KS * 16; ++k) {
                const int cell = pair_cell[c][k < (int)pair_cell[c].size() ? k : 0];
                k16[(size_t)c * KS * 16 + k] = (unsigned short)((cell / W) * pitch + cell % W);
            }
    }
    ints.insert(ints.end(), pos_angle.begin(), pos_angle.end());
    void *d_atab = nullptr, *d_koff = nullptr, *d_angles = nullptr, *d_flags = nullptr;
    TOPO_TRY(upload_table(4, atab.data(), atab.size() * sizeof(uint16_t), &d_atab));
    TOPO_TRY(upload_table(5, ints.data(), ints.size() * sizeof(int), &d_koff));
    TOPO_TRY(upload_table(2, angles, (size_t)n_angles * sizeof(float), &d_angles));
    dim3 grid((b.nx + kTW - 1) / kTW, (b.out_rows + kTH - 1) / kTH);
    const size_t flag_bytes = (size_t)grid.x * grid.y * sizeof(int);
    TOPO_TRY(workspace(3, flag_bytes, &d_flags));
    TOPO_HIP(hipMemsetAsync(d_flags, 0, flag_bytes, ctx().compute));
    VsArgs a{};
    a.in = b.in;
    a.norm = norm_out;
    a.dir = dir_out;
    a.atab = (const unsigned char*)d_atab;
    a.koff = (const unsigned short*)d_koff;
    a.pair_sum[0] = centre[0] * pitch + centre[0];
    a.pair_sum[1] = n_cls > 1 ? centre[1] * pitch + centre[1] : 0;
    a.pos_angle = (const int*)d_koff + koff_ints;
    a.angles = (const float*)d_angles;
    a.flags = (int*)d_flags;
    a.class_tiles[0] = class_tiles[0];
    a.class_tiles[1] = class_tiles[1];
    a.ks = KS;
    a.w = W;
    a.pitch = pitch;
    a.in_rows = b.in_rows;
    a.in_row0 = b.in_row0;
    a.gny = b.gny;
    a.nx = b.nx;
    a.out_row0 = b.out_row0;
    a.out_rows = b.out_rows;
    a.mean = (float)mean;
    a.stdev = (float)stdev;
    *flags_out = (const int*)d_flags;
    *flag_cols = (int)grid.x;
    *folded = 2;
    switch (n_planes) {
        case 1: return launch_stream_np<1>(a, grid);
        case 2: return launch_stream_np<2>(a, grid);
        case 3: return launch_stream_np<3>(a, grid);
        default: return launch_stream_np<4>(a, grid);
    }
}

// The folded form, if the tables allow it: every kernel bit for bit point-symmetric, at most two canvas centres in the common
// window, at most 15 K steps of live pairs (kernels of up to 17 px).  *folded = 0: not such a case, nothing launched.
int launch_fold(const Block& b, const float* taps, const int32_t* ksize, const float* angles, int n_angles, int n_planes, int W,
                double mean, double stdev, float* norm_out, float* dir_out, const int** flags_out, int* flag_cols, int* folded) {
    *folded = 0;
    // symmetry, and the class of every angle: the window cell opposite (wy, wx) is (C - wy, C - wx), C = ks - 1 + 2 (W / 2 - ks / 2)
    std::vector<int> cls_of(n_angles), centre;
    const float* src = taps;
    for (int ang = 0; ang < n_angles; ++ang) {
        const int ks = ksize[ang];
        for (int ky = 0; ky < ks; ++ky)
            for (int kx = 0; kx < ks; ++kx)
                for (int q = 0; q < n_planes; ++q)
                    if (src[((size_t)ky * ks + kx) * 4 + q] != src[((size_t)(ks - 1 - ky) * ks + (ks - 1 - kx)) * 4 + q]) return TOPO_AMD_OK;
        const int C = ks - 1 + 2 * (W / 2 - ks / 2);
        size_t k = 0;
        while (k < centre.size() && centre[k] != C) ++k;
        if (k == centre.size()) centre.push_back(C);
        if (centre.size() > 2) return TOPO_AMD_OK;
        cls_of[ang] = (int)k;
        src += (size_t)ks * ks * 4;
    }
    const int n_cls = (int)centre.size();
    // the live pairs of each class (a pair by its first cell in row-major order), and each angle's taps on its class's pairs
    std::vector<int> pair_of[2];     // window cell -> pair index of the class, or -1
    std::vector<int> pair_cell[2];   // pair -> its first cell
    for (int c = 0; c < n_cls; ++c) pair_of[c].assign((size_t)W * W, -1);
    src = taps;
    for (int ang = 0; ang < n_angles; ++ang) {
        const int ks = ksize[ang], sh = W / 2 - ks / 2, c = cls_of[ang], C = centre[c];
        for (int ky = 0; ky < ks; ++ky)
            for (int kx = 0; kx < ks; ++kx) {
                bool any = false;
                for (int q = 0; q < n_planes; ++q) any = any || src[((size_t)ky * ks + kx) * 4 + q] != 0.0f;
                if (!any) continue;
                const int wy = ky + sh, wx = kx + sh, oy = C - wy, ox = C - wx;
                if (oy < 0 || oy >= W || ox < 0 || ox >= W) return TOPO_AMD_OK;  // (cannot happen: the partner is a cell of the same canvas)
                const int cell = std::min(wy * W + wx, oy * W + ox);
                if (pair_of[c][cell] < 0) pair_of[c][cell] = -2;  // live; numbered below in row-major order
            }
        src += (size_t)ks * ks * 4;
    }
    int KS = 0;
    for (int c = 0; c < n_cls; ++c) {
        for (int cell = 0; cell < W * W; ++cell)
            if (pair_of[c][cell] == -2) {
                pair_of[c][cell] = (int)pair_cell[c].size();
                pair_cell[c].push_back(cell);
            }
        KS = std::max(KS, ((int)pair_cell[c].size() + 15) / 16);
    }
    if (KS < 1) return TOPO_AMD_OK;
    if (KS > kMaxSteps || W > kValleyMfmaMaxKernel)
        return launch_fold_stream(b, taps, ksize, angles, n_angles, n_planes, W, mean, stdev, norm_out, dir_out, flags_out, flag_cols,
                                  folded, n_cls, cls_of, centre, pair_of, pair_cell);
    const int aph = 16 / n_planes, apt = 2 * aph, gt = group_tiles(KS);
    // the stream: the tiles of class 0 (its angles in rising order, the last group filled with copies of its last angle), then class 1
    std::vector<int> pos_angle;
    int class_groups[2] = {0, 0}, class_tiles[2] = {0, 0};
    for (int c = 0; c < n_cls; ++c) {
        std::vector<int> mine;
        for (int ang = 0; ang < n_angles; ++ang)
            if (cls_of[ang] == c) mine.push_back(ang);
        const int tiles = ((int)mine.size() + apt - 1) / apt;
        class_tiles[c] = tiles;
        class_groups[c] = (tiles + gt - 1) / gt;
        for (int k = 0; k < class_groups[c] * gt * apt; ++k) pos_angle.push_back(mine[std::min(k, (int)mine.size() - 1)]);
    }
    const int n_groups = class_groups[0] + class_groups[1];
    const size_t frag_halfs = kFragBytes / 2;
    std::vector<uint16_t> atab((size_t)n_groups * gt * KS * 2 * frag_halfs, 0);
    std::vector<size_t> first_tap((size_t)n_angles);
    {
        size_t at = 0;
        for (int ang = 0; ang < n_angles; ++ang) {
            first_tap[ang] = at;
            at += (size_t)ksize[ang] * ksize[ang] * 4;
        }
    }
    for (size_t pos = 0; pos < pos_angle.size(); ++pos) {
        const int ang = pos_angle[pos], c = cls_of[ang], C = centre[c];
        const int ks = ksize[ang], sh = W / 2 - ks / 2;
        const int tile = (int)pos / apt, hrow = ((int)pos % apt) / aph, slot = ((int)pos % apt) % aph;
        const float* t = taps + first_tap[ang];
        for (int ky = 0; ky < ks; ++ky)
            for (int kx = 0; kx < ks; ++kx) {
                const int wy = ky + sh, wx = kx + sh, cell = wy * W + wx, other = (C - wy) * W + (C - wx);
                if (cell > other) continue;  // the pair's second cell: the same tap
                const int k = pair_of[c][cell];
                if (k < 0) continue;
                for (int q = 0; q < n_planes; ++q) {
                    float v = t[((size_t)ky * ks + kx) * 4 + q];
                    if (v == 0.0f) continue;
                    if (cell == other) v *= 0.5f;  // the centre cell is its own partner: z + z against half the tap (exact)
                    const uint16_t hi = f16_bits(v);
                    const uint16_t lo = f16_bits(v - f16_value(hi));
                    const int r = slot * n_planes + q, m = 8 * (r / 4) + 4 * hrow + r % 4;
                    const int s = k / 16, kh = (k % 16) / 8, e = k % 8;
                    const size_t frag = ((size_t)tile * KS + s) * 2;
                    const size_t at = (size_t)(m + 32 * kh) * 8 + e;
                    atab[frag * frag_halfs + at] = hi;
                    atab[(frag + 1) * frag_halfs + at] = lo;
                }
            }
    }
    std::vector<int> koff((size_t)2 * KS * 16 * 2, 0);
    for (int c = 0; c < n_cls; ++c)
        for (int k = 0; k < KS * 16; ++k) {
            const int cell = pair_cell[c][k < (int)pair_cell[c].size() ? k : 0];  // behind the last live pair: the first one, zero taps
            const int wy = cell / W, wx = cell % W;
            koff[((size_t)c * KS * 16 + k) * 2] = wy * kPitch + wx;
            koff[((size_t)c * KS * 16 + k) * 2 + 1] = (centre[c] - wy) * kPitch + (centre[c] - wx);
        }
    void *d_atab = nullptr, *d_koff = nullptr, *d_angles = nullptr, *d_flags = nullptr, *d_pos = nullptr;
    // one table: the offsets, then the positions' angles (the table slots are few)
    std::vector<int> ints(koff);
    ints.insert(ints.end(), pos_angle.begin(), pos_angle.end());
    TOPO_TRY(upload_table(4, atab.data(), atab.size() * sizeof(uint16_t), &d_atab));
    TOPO_TRY(upload_table(5, ints.data(), ints.size() * sizeof(int), &d_koff));
    d_pos = (char*)d_koff + koff.size() * sizeof(int);
    TOPO_TRY(upload_table(2, angles, (size_t)n_angles * sizeof(float), &d_angles));
    dim3 grid((b.nx + kTW - 1) / kTW, (b.out_rows + kTH - 1) / kTH);
    const size_t flag_bytes = (size_t)grid.x * grid.y * sizeof(int);
    TOPO_TRY(workspace(3, flag_bytes, &d_flags));
    TOPO_HIP(hipMemsetAsync(d_flags, 0, flag_bytes, ctx().compute));
    VfArgs a{};
    a.in = b.in;
    a.norm = norm_out;
    a.dir = dir_out;
    a.atab = (const unsigned char*)d_atab;
    a.koff = (const int*)d_koff;
    a.pos_angle = (const int*)d_pos;
    a.angles = (const float*)d_angles;
    a.flags = (int*)d_flags;
    a.class_groups[0] = class_groups[0];
    a.class_groups[1] = class_groups[1];
    a.class_tiles[0] = class_tiles[0];
    a.class_tiles[1] = class_tiles[1];
    a.w = W;
    a.in_rows = b.in_rows;
    a.in_row0 = b.in_row0;
    a.gny = b.gny;
    a.nx = b.nx;
    a.out_row0 = b.out_row0;
    a.out_rows = b.out_rows;
    a.mean = (float)mean;
    a.stdev = (float)stdev;
    *flags_out = (const int*)d_flags;
    *flag_cols = (int)grid.x;
    *folded = 1;
    switch (KS) {
#define TOPO_KS(k) case k: return launch_fold_np<k>(a, grid, n_planes);
        TOPO_KS(1) TOPO_KS(2) TOPO_KS(3) TOPO_KS(4) TOPO_KS(5) TOPO_KS(6) TOPO_KS(7) TOPO_KS(8) TOPO_KS(9) TOPO_KS(10)
        TOPO_KS(11) TOPO_KS(12) TOPO_KS(13) TOPO_KS(14) TOPO_KS(15)
#undef TOPO_KS
    }
    return TOPO_AMD_OK;
}

}  // namespace

// The matrix-pipe evaluation over the block.  *done = 0: not a case for it (no live cell, or more than 240), nothing
// launched; 1: the kernel over the live cells; 2: the folded form (point-symmetric tables).  Otherwise *flags_out / *flag_cols describe the tiles (kValleyMfmaTileRows rows x 64 columns, anchored at
// out_row0) in which it left pixels marked norm = -1 for the direct kernel.
int launch_valley_ridge_mfma(const Block& b, const float* taps, const int32_t* ksize, const float* angles, int n_angles,
                             int n_planes, int kmax, double mean, double stdev, float* norm_out, float* dir_out,
                             const int** flags_out, int* flag_cols, int* done) {
    *done = 0;
    TOPO_REQUIRE(kmax >= 1 && kmax <= kValleyStreamMaxKernel, "valley_ridge (matrix pipe): kernel side %d", kmax);
    const int W = kmax;  // the window = the largest canvas (a smaller one sits inside it: it starts ks / 2 before the pixel)
    {
        const char* e = std::getenv("TOPO_AMD_VALLEY_FOLD");  // 0: never the folded form (read at every launch; tests)
        if (!(e && *e == '0')) {
            int folded = 0;
            TOPO_TRY(launch_fold(b, taps, ksize, angles, n_angles, n_planes, W, mean, stdev, norm_out, dir_out, flags_out, flag_cols,
                                 &folded));
            if (folded) {
                *done = 1 + folded;  // 2: the operands in registers, 3: streamed
                return TOPO_AMD_OK;
            }
        }
    }
    if (W > kValleyMfmaMaxKernel) return TOPO_AMD_OK;  // (the form over the cells keeps a pixel tile's operands in registers)
    // the canvases of all (angle, plane) filters in the common window, and the cells in which any of them has a tap
    std::vector<float> canvas((size_t)n_angles * n_planes * W * W, 0.0f);
    std::vector<char> live_cell((size_t)W * W, 0);
    const float* src = taps;
    for (int ang = 0; ang < n_angles; ++ang) {
        const int ks = ksize[ang];
        const int shift = W / 2 - ks / 2;
        for (int ky = 0; ky < ks; ++ky)
            for (int kx = 0; kx < ks; ++kx)
                for (int q = 0; q < n_planes; ++q) {
                    const float t = src[((size_t)ky * ks + kx) * 4 + q];
                    if (t == 0.0f) continue;
                    const size_t cell = (size_t)(ky + shift) * W + kx + shift;
                    canvas[((size_t)ang * n_planes + q) * W * W + cell] = t;
                    live_cell[cell] = 1;
                }
        src += (size_t)ks * ks * 4;
    }
    std::vector<int> live;
    for (int c = 0; c < W * W; ++c)
        if (live_cell[c]) live.push_back(c);
    const int KS = ((int)live.size() + 15) / 16;
    if (KS < 1 || KS > kMaxSteps) return TOPO_AMD_OK;
    std::vector<int> koff((size_t)KS * 16);
    for (int k = 0; k < KS * 16; ++k) {
        const int c = live[k < (int)live.size() ? k : 0];  // behind the last live cell: the first one again, against zero taps
        koff[k] = (c / W) * kPitch + c % W;
    }
    const int aph = 16 / n_planes, apt = 2 * aph;
    const int n_tiles = (n_angles + apt - 1) / apt;
    const int gt = group_tiles(KS);
    const int n_groups = (n_tiles + gt - 1) / gt;
    const size_t frag_halfs = kFragBytes / 2;
    std::vector<uint16_t> atab((size_t)n_groups * gt * KS * 2 * frag_halfs, 0);
    // the last group of tiles is filled up with copies of the last angle: a copy never beats the original under the strict comparison
    for (int ang = 0; ang < n_groups * gt * apt; ++ang) {
        const int tile = ang / apt, hrow = (ang % apt) / aph, slot = (ang % apt) % aph;
        for (int q = 0; q < n_planes; ++q) {
            const float* cv = canvas.data() + ((size_t)std::min(ang, n_angles - 1) * n_planes + q) * W * W;
            const int v = slot * n_planes + q;             // the result register of the lane half
            const int m = 8 * (v / 4) + 4 * hrow + v % 4;  // its row of the filter tile
            for (int k = 0; k < (int)live.size(); ++k) {
                const float t = cv[live[k]];
                if (t == 0.0f) continue;
                const uint16_t hi = f16_bits(t);
                const uint16_t lo = f16_bits(t - f16_value(hi));
                const int s = k / 16, kh = (k % 16) / 8, e = k % 8;  // K step, K half = lane half of the A operand, element
                const size_t frag = ((size_t)tile * KS + s) * 2;
                const size_t at = (size_t)(m + 32 * kh) * 8 + e;
                atab[frag * frag_halfs + at] = hi;
                atab[(frag + 1) * frag_halfs + at] = lo;
            }
        }
    }
    void *d_atab = nullptr, *d_koff = nullptr, *d_angles = nullptr, *d_flags = nullptr;
    TOPO_TRY(upload_table(4, atab.data(), atab.size() * sizeof(uint16_t), &d_atab));
    TOPO_TRY(upload_table(5, koff.data(), koff.size() * sizeof(int), &d_koff));
    TOPO_TRY(upload_table(2, angles, (size_t)n_angles * sizeof(float), &d_angles));
    dim3 grid((b.nx + kTW - 1) / kTW, (b.out_rows + kTH - 1) / kTH);
    const size_t flag_bytes = (size_t)grid.x * grid.y * sizeof(int);
    TOPO_TRY(workspace(3, flag_bytes, &d_flags));
    TOPO_HIP(hipMemsetAsync(d_flags, 0, flag_bytes, ctx().compute));
    VmArgs a{};
    a.in = b.in;
    a.norm = norm_out;
    a.dir = dir_out;
    a.atab = (const unsigned char*)d_atab;
    a.koff = (const int*)d_koff;
    a.angles = (const float*)d_angles;
    a.flags = (int*)d_flags;
    a.n_angles = n_angles;
    a.n_groups = n_groups;
    a.w = W;
    a.in_rows = b.in_rows;
    a.in_row0 = b.in_row0;
    a.gny = b.gny;
    a.nx = b.nx;
    a.out_row0 = b.out_row0;
    a.out_rows = b.out_rows;
    a.mean = (float)mean;
    a.stdev = (float)stdev;
    *flags_out = (const int*)d_flags;
    *flag_cols = (int)grid.x;
    *done = 1;
    switch (KS) {
#define TOPO_KS(k) case k: return launch_np<k>(a, grid, n_planes);
        TOPO_KS(1) TOPO_KS(2) TOPO_KS(3) TOPO_KS(4) TOPO_KS(5) TOPO_KS(6) TOPO_KS(7) TOPO_KS(8) TOPO_KS(9) TOPO_KS(10)
        TOPO_KS(11) TOPO_KS(12) TOPO_KS(13) TOPO_KS(14) TOPO_KS(15)
#undef TOPO_KS
    }
    return TOPO_AMD_OK;
}

}  // namespace topo
