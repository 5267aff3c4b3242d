// Compile-time geometry of the reference's circular_kernel(size) (topo.py:191-213) as seen by
// scipy.signal.convolve(..., mode="same") (topo.py:175): for every column offset di the set
// taps form ONE vertical run dj in [lo(di), hi(di)].  The mask is symmetric under transposition,
// so the same table describes the horizontal runs per row offset.
//
// Used by the wave-shift disc kernel (disc_wave.hip), which needs every table entry as a
// compile-time constant so that the per-run column sums live in statically indexed registers.
#pragma once

namespace topo {

template <int SIZE>
struct DiscTable {
    int off_min = 0, off_max = 0;  // offsets covered, both axes
    int centre = 0;                // offset of the tap TPI zeroes (0 for odd sizes, -1 for even)
    int taps = 0;
    int lo[SIZE] = {};             // per offset index (di - off_min): run [lo, hi] of dj
    int hi[SIZE] = {};
    int run_of[SIZE] = {};         // index of the distinct (lo, hi) pair
    int num_runs = 0;
    int run_lo[SIZE] = {};         // the distinct pairs
    int run_hi[SIZE] = {};
    int colpre[SIZE + 1] = {};     // taps in the columns before offset index k (colpre[SIZE] = taps)
};

template <int SIZE>
constexpr DiscTable<SIZE> make_disc_table() {
    DiscTable<SIZE> t;
    const int m = SIZE / 2;        // int(size / 2), topo.py:205
    const int c = (SIZE - 1) / 2;  // first kept index of mode="same"
    t.off_min = c - (SIZE - 1);
    t.off_max = c;
    t.centre = c - m;
    for (int b = 0; b < SIZE; ++b) {
        int a_lo = SIZE, a_hi = -1;
        for (int a = 0; a < SIZE; ++a) {
            const bool on = SIZE < 5 || ((a - m) * (a - m) + (b - m) * (b - m) <= m * m);
            if (on) {
                ++t.taps;
                if (a < a_lo) a_lo = a;
                if (a > a_hi) a_hi = a;
            }
        }
        // kernel column b is offset di = c - b; kernel rows [a_lo, a_hi] are dj in [c-a_hi, c-a_lo]
        const int idx = (c - b) - t.off_min;
        t.lo[idx] = c - a_hi;
        t.hi[idx] = c - a_lo;
    }
    for (int i = 0; i < SIZE; ++i) {
        int found = -1;
        for (int r = 0; r < t.num_runs; ++r)
            if (t.run_lo[r] == t.lo[i] && t.run_hi[r] == t.hi[i]) found = r;
        if (found < 0) {
            found = t.num_runs++;
            t.run_lo[found] = t.lo[i];
            t.run_hi[found] = t.hi[i];
        }
        t.run_of[i] = found;
    }
    for (int i = 0; i < SIZE; ++i) t.colpre[i + 1] = t.colpre[i] + (t.hi[i] - t.lo[i] + 1);
    return t;
}

}  // namespace topo
