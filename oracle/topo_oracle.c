/* C / OpenMP twin of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see oracle/topo_oracle.py): used
 * by tests and by bench.py's cpu_baseline leg as the "all host cores" CPU baseline BASELINE.md
 * section 4 asks for, and as the only CPU baseline for Sx (the reference's Sx needs numba).
 *
 * It evaluates the reference's formulas directly in float64:
 *   - TPI / STD over the disc of circular_kernel(size) with mode="same" zero padding
 *     (reference topo.py:168-181, :295-307, :205-213), via per-row prefix sums;
 *   - Sx: max over the offset table of the elevation angle (reference topo.py:928-953).
 * Checked against oracle/topo_oracle.py (which is pinned to the reference's golden vectors) in
 * tests/test_oracle_c_twin.py.
 *
 * Build: gcc -O3 -fopenmp -shared -fPIC -o oracle/libtopo_oracle.so oracle/topo_oracle.c -lm
 */
#include <math.h>
#include <stdlib.h>
#include <omp.h>

static int disc_on(int size, int a, int b) {
    const int m = size / 2;
    if (size < 5) return 1;
    return (a - m) * (a - m) + (b - m) * (b - m) <= m * m;
}

/* tpi and/or sd may be NULL; both are float64 ny x nx */
int oracle_tpi_std(const float* dem, int ny, int nx, int size, double* tpi, double* sd) {
    const int c = (size - 1) / 2, m = size / 2;
    int* lo = (int*)malloc(sizeof(int) * size);
    int* hi = (int*)malloc(sizeof(int) * size);
    long taps = 0;
    for (int a = 0; a < size; ++a) { /* kernel row a <-> offset dj = c - a; run of di */
        int b0 = size, b1 = -1;
        for (int b = 0; b < size; ++b)
            if (disc_on(size, a, b)) { if (b < b0) b0 = b; if (b > b1) b1 = b; ++taps; }
        lo[a] = c - b1;
        hi[a] = c - b0;
    }
    const double n = (double)taps;
    /* row prefix sums of x and trunc(x)^2, exclusive, nx + 1 entries per row */
    double* p1 = (double*)malloc(sizeof(double) * (size_t)ny * (nx + 1));
    /* (the plane of trunc(x)^2 only when STD is wanted: TPI alone then touches half the memory) */
    double* p2 = sd ? (double*)malloc(sizeof(double) * (size_t)ny * (nx + 1)) : NULL;
    if (!lo || !hi || !p1 || (sd && !p2)) return -1;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ny; ++j) {
        double s1 = 0.0, s2 = 0.0;
        double* q1 = p1 + (size_t)j * (nx + 1);
        double* q2 = p2 ? p2 + (size_t)j * (nx + 1) : NULL;
        q1[0] = 0.0;
        if (q2) q2[0] = 0.0;
        for (int i = 0; i < nx; ++i) {
            const double x = dem[(size_t)j * nx + i], t = trunc(x);
            s1 += x;
            s2 += t * t;
            q1[i + 1] = s1;
            if (q2) q2[i + 1] = s2;
        }
    }
#pragma omp parallel for schedule(dynamic, 8)
    for (int j = 0; j < ny; ++j) {
        for (int i = 0; i < nx; ++i) {
            double s1 = 0.0, s2 = 0.0;
            for (int a = 0; a < size; ++a) {
                const int y = j + c - a;
                if (y < 0 || y >= ny) continue;
                int x0 = i + lo[a], x1 = i + hi[a];
                if (x0 < 0) x0 = 0;
                if (x1 > nx - 1) x1 = nx - 1;
                if (x1 < x0) continue;
                s1 += p1[(size_t)y * (nx + 1) + x1 + 1] - p1[(size_t)y * (nx + 1) + x0];
                if (p2) s2 += p2[(size_t)y * (nx + 1) + x1 + 1] - p2[(size_t)y * (nx + 1) + x0];
            }
            const size_t o = (size_t)j * nx + i;
            if (tpi) {
                const int cy = j + c - m, cx = i + c - m; /* the tap TPI zeroes */
                const double xc = (cy >= 0 && cy < ny && cx >= 0 && cx < nx) ? dem[(size_t)cy * nx + cx] : 0.0;
                tpi[o] = (double)dem[o] - (s1 - xc) / (n - 1.0);
            }
            if (sd) {
                double var = (s2 - s1 * s1 / n) / (n - 1.0);
                if (var < 0.0) var = 0.0;
                sd[o] = sqrt(var);
            }
        }
    }
    free(lo); free(hi); free(p1); free(p2);
    return 0;
}

/* offsets (dj, di) with distances; NaN distance = skip; frame of `window` pixels stays 0 */
int oracle_sx(const float* dem, int ny, int nx, const int* dj, const int* di, const double* dist,
              int n_off, int window, double height, float* out) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ny; ++j) {
        for (int i = 0; i < nx; ++i) {
            const size_t o = (size_t)j * nx + i;
            if (j < window || j >= ny - window || i < window || i >= nx - window) { out[o] = 0.0f; continue; }
            const double centre = (double)dem[o] + height;
            double best = -INFINITY;
            int any = 0;
            for (int k = 0; k < n_off; ++k) {
                if (isnan(dist[k])) continue;
                const double ang = atan(((double)dem[(size_t)(j + dj[k]) * nx + i + di[k]] - centre) / dist[k]);
                if (isnan(ang)) continue;
                if (ang > best) best = ang;
                any = 1;
            }
            out[o] = any ? (float)(best * (180.0 / M_PI)) : NAN;
        }
    }
    return 0;
}

int oracle_threads(void) { return omp_get_max_threads(); }
