/* C / OpenMP twin of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see oracle/topo_oracle.py): used
 * by tests and by bench.py's cpu_baseline leg as the "all host cores" CPU baseline BASELINE.md
 * section 4 asks for, and as the only CPU baseline for Sx (the reference's Sx needs numba).
 *
 * It evaluates the reference's formulas directly in float64:
 *   - TPI / STD over the disc of circular_kernel(size) with mode="same" zero padding
 *     (reference topo.py:168-181, :295-307, :205-213), via per-row prefix sums, tile by tile;
 *   - Sx: max over the offset table of the elevation angle (reference topo.py:928-953).
 * Checked against oracle/topo_oracle.py (which is pinned to the reference's golden vectors) in
 * tests/test_oracle_c_twin.py.
 *
 * Build (oracle/build_oracle.py): gcc -O3 -mavx2 -mfma -fopenmp -shared -fPIC -o oracle/libtopo_oracle.so oracle/topo_oracle.c -lm
 * (AVX2, not -march=native: the library is built in one container and timed on another host)
 */
#include <math.h>
#include <stdlib.h>
#include <omp.h>

static int disc_on(int size, int a, int b) {
    const int m = size / 2;
    if (size < 5) return 1;
    return (a - m) * (a - m) + (b - m) * (b - m) <= m * m;
}

/* tpi and/or sd may be NULL; both are float64 ny x nx.
 * Work is cut into tiles of TY x TX pixels, one tile per task.  A thread forms the row prefix sums of its tile plus
 * the disc's halo in a private buffer (float64, running from the tile's own first column: 0.6 MB, resident in its L2)
 * and then, for every output row, walks the disc's rows with the pixel loop innermost: acc[i] += P[y][i + hi + 1] -
 * P[y][i + lo] over contiguous i, which the compiler vectorises.  (The first version kept whole-DEM prefix planes and
 * looped over the disc rows per pixel: 134 loads per pixel from 67 rows 128 KB apart, and 256 threads were no faster
 * than one core of scipy's FFT.) */
enum { TY = 64, TX = 512 };

int oracle_tpi_std(const float* dem, int ny, int nx, int size, double* tpi, double* sd) {
    const int c = (size - 1) / 2, m = size / 2;
    int* lo = (int*)malloc(sizeof(int) * size);
    int* hi = (int*)malloc(sizeof(int) * size);
    if (!lo || !hi) return -1;
    long taps = 0;
    int reach = 0; /* largest |di| */
    for (int a = 0; a < size; ++a) { /* kernel row a <-> offset dj = c - a; run of di */
        int b0 = size, b1 = -1;
        for (int b = 0; b < size; ++b)
            if (disc_on(size, a, b)) { if (b < b0) b0 = b; if (b > b1) b1 = b; ++taps; }
        lo[a] = c - b1;
        hi[a] = c - b0;
        if (b1 >= b0) {
            if (abs(lo[a]) > reach) reach = abs(lo[a]);
            if (abs(hi[a]) > reach) reach = abs(hi[a]);
        }
    }
    const double n = (double)taps;
    const int tiles_y = (ny + TY - 1) / TY, tiles_x = (nx + TX - 1) / TX;
    const int rows_b = TY + size, cols_b = TX + 2 * reach + 2; /* buffer: tile + halo, one extra prefix column */
    int failed = 0;
#pragma omp parallel
    {
        double* p1 = (double*)malloc(sizeof(double) * (size_t)rows_b * cols_b);
        double* p2 = sd ? (double*)malloc(sizeof(double) * (size_t)rows_b * cols_b) : NULL;
        double* a1 = (double*)malloc(sizeof(double) * TX);
        double* a2 = (double*)malloc(sizeof(double) * TX);
        if (!p1 || (sd && !p2) || !a1 || !a2) {
#pragma omp atomic write
            failed = 1;
        } else {
#pragma omp for schedule(dynamic, 1) collapse(2)
            for (int ty = 0; ty < tiles_y; ++ty) {
                for (int tx = 0; tx < tiles_x; ++tx) {
                    const int j0 = ty * TY, i0 = tx * TX;
                    const int th = (j0 + TY <= ny ? TY : ny - j0), tw = (i0 + TX <= nx ? TX : nx - i0);
                    const int y_first = j0 + c - (size - 1), x_first = i0 - reach; /* DEM row / column of buffer (0, 0) */
                    const int nrow = th + size - 1, ncol = tw + 2 * reach;
                    /* exclusive row prefix sums; samples outside the DEM are the zero padding of mode="same" */
                    for (int r = 0; r < nrow; ++r) {
                        const int y = y_first + r;
                        double* q1 = p1 + (size_t)r * cols_b;
                        double* q2 = p2 ? p2 + (size_t)r * cols_b : NULL;
                        double s1 = 0.0, s2 = 0.0;
                        q1[0] = 0.0;
                        if (q2) q2[0] = 0.0;
                        for (int k = 0; k < ncol; ++k) {
                            const int x = x_first + k;
                            double v = 0.0, t = 0.0;
                            if (y >= 0 && y < ny && x >= 0 && x < nx) { v = dem[(size_t)y * nx + x]; t = trunc(v); }
                            s1 += v;
                            s2 += t * t;
                            q1[k + 1] = s1;
                            if (q2) q2[k + 1] = s2;
                        }
                    }
                    for (int jj = 0; jj < th; ++jj) {
                        const int j = j0 + jj;
                        for (int i = 0; i < tw; ++i) { a1[i] = 0.0; a2[i] = 0.0; }
                        for (int a = 0; a < size; ++a) {
                            if (hi[a] < lo[a]) continue;
                            const int r = (j + c - a) - y_first; /* always inside the buffer */
                            const double* e1 = p1 + (size_t)r * cols_b + (reach + hi[a] + 1);
                            const double* b1 = p1 + (size_t)r * cols_b + (reach + lo[a]);
                            for (int i = 0; i < tw; ++i) a1[i] += e1[i] - b1[i];
                            if (p2) {
                                const double* e2 = p2 + (size_t)r * cols_b + (reach + hi[a] + 1);
                                const double* b2 = p2 + (size_t)r * cols_b + (reach + lo[a]);
                                for (int i = 0; i < tw; ++i) a2[i] += e2[i] - b2[i];
                            }
                        }
                        for (int i = 0; i < tw; ++i) {
                            const size_t o = (size_t)j * nx + i0 + i;
                            if (tpi) {
                                const int cy = j + c - m, cx = i0 + i + c - m; /* the tap TPI zeroes */
                                const double xc = (cy >= 0 && cy < ny && cx >= 0 && cx < nx) ? dem[(size_t)cy * nx + cx] : 0.0;
                                tpi[o] = (double)dem[o] - (a1[i] - xc) / (n - 1.0);
                            }
                            if (sd) {
                                double var = (a2[i] - a1[i] * a1[i] / n) / (n - 1.0);
                                if (var < 0.0) var = 0.0;
                                sd[o] = sqrt(var);
                            }
                        }
                    }
                }
            }
        }
        free(p1); free(p2); free(a1); free(a2);
    }
    free(lo); free(hi);
    return failed ? -1 : 0;
}

/* offsets (dj, di) with distances; NaN distance = skip; frame of `window` pixels stays 0 */
int oracle_sx(const float* dem, int ny, int nx, const int* dj, const int* di, const double* dist,
              int n_off, int window, double height, float* out) {
    /* the rays of a sector share most of their pixels (240 points, 32 distinct at 500 m / 30 m): a maximum does not
     * care about repeats, so each distinct (dj, di, dist) is looked at once */
    int* udj = (int*)malloc(sizeof(int) * (n_off > 0 ? n_off : 1));
    int* udi = (int*)malloc(sizeof(int) * (n_off > 0 ? n_off : 1));
    double* udist = (double*)malloc(sizeof(double) * (n_off > 0 ? n_off : 1));
    if (!udj || !udi || !udist) return -1;
    int nu = 0;
    for (int k = 0; k < n_off; ++k) {
        if (isnan(dist[k])) continue;
        int seen = 0;
        for (int q = 0; q < nu && !seen; ++q) seen = udj[q] == dj[k] && udi[q] == di[k] && udist[q] == dist[k];
        if (!seen) { udj[nu] = dj[k]; udi[nu] = di[k]; udist[nu] = dist[k]; ++nu; }
    }
    dj = udj; di = udi; dist = udist; n_off = nu;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ny; ++j) {
        for (int i = 0; i < nx; ++i) {
            const size_t o = (size_t)j * nx + i;
            if (j < window || j >= ny - window || i < window || i >= nx - window) { out[o] = 0.0f; continue; }
            const double centre = (double)dem[o] + height;
            /* atan is monotone: the largest angle belongs to the largest tangent, one atan per pixel (the reference
             * takes one per ray point, topo.py:948; the first version of this twin did too and 256 threads managed
             * 5 Mpixels/s) */
            double best = -INFINITY;
            int any = 0;
            for (int k = 0; k < n_off; ++k) {
                if (isnan(dist[k])) continue;
                const double tangent = ((double)dem[(size_t)(j + dj[k]) * nx + i + di[k]] - centre) / dist[k];
                if (isnan(tangent)) continue;
                if (tangent > best) best = tangent;
                any = 1;
            }
            best = atan(best);
            out[o] = any ? (float)(best * (180.0 / M_PI)) : NAN;
        }
    }
    free(udj); free(udi); free(udist);
    return 0;
}

int oracle_threads(void) { return omp_get_max_threads(); }
