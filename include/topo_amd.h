/*
 * topo_amd.h - C ABI of libtopo_amd.so, the MI355X (gfx950) implementation of the
 * per-pixel descriptor hot path of MeteoSwiss/topo-descriptors.
 *
 * The reference has no FFI of its own (it is pure Python on scipy/numba wheels); the
 * boundary this library replaces is the set of Python call sites listed next to each
 * entry point (file:line into the reference).  INTEGRATION.md shows the ctypes stub a
 * maintainer adds to `topo_descriptors/topo.py` to route those calls here.
 *
 * Conventions
 *   - every function returns 0 on success, a negative TOPO_AMD_E* code otherwise; the
 *     message is available from topo_amd_last_error() (thread-local).
 *   - arrays are C-contiguous float32, ny rows x nx columns, row pitch == nx.
 *   - *_f32 entry points take HOST pointers, do upload -> kernels -> download and return
 *     when the result is in the caller's buffer.  No pointer is retained after return (the
 *     device planes are: topo_amd_release_host_planes).
 *   - *_dev entry points take DEVICE pointers obtained from topo_amd_malloc, enqueue on the
 *     library's compute stream and return immediately; call topo_amd_sync() before reading.
 *   - row-block form: a device block holds `in_rows` consecutive rows of a global
 *     `gny x nx` DEM starting at global row `in_row0`.  Output rows are
 *     [out_row0, out_row0+out_rows) in global numbering and `out` points at the first of
 *     them.  The boundary rule of each descriptor (zero padding, reflection, one-sided
 *     difference, zero frame) is applied at the GLOBAL edges only, so that a row shard
 *     plus its ghost rows gives results bit-identical to the single-block run.  The block
 *     must contain every row the requested output rows depend on (see topo_amd_halo_rows).
 *   - one process drives one GPU (topo_amd_init(device)); entry points are serialised on
 *     that device's compute stream.
 *
 * Threading
 *   Every entry point may be called from any thread at any time.  The context (workspaces, the device planes of the
 *   host-buffer entry points, the streams and events of their pipeline, the gate of a sharded call) is guarded by ONE
 *   mutex that every entry point touching it holds from its first to its last line, so concurrent calls run one after
 *   the other, each exactly as if it were alone - same bits as a serial run.  A host-buffer call (*_f32) holds it until its
 *   result is in the caller's array; a device call (*_dev, topo_amd_shard_*) only while it enqueues, so the GPU work of
 *   several threads queues up back to back on the compute stream (in the order the threads took the mutex).
 *   topo_amd_sync() waits for everything enqueued so far, by whichever thread.  The calling thread is bound to the
 *   context's device on entry (hipSetDevice).  topo_amd_last_error() and topo_amd_shard_layout are per thread; the
 *   stopwatch (topo_amd_timer_*, topo_amd_mark*) is one per process and means what it says only when one thread launches
 *   between its two ends.  Collective calls (topo_amd_shard_*, topo_amd_comm_*) must be issued in the same order on
 *   every rank: drive them from one thread per process.
 */
#ifndef TOPO_AMD_H
#define TOPO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TOPO_AMD_OK 0
#define TOPO_AMD_EINVAL (-1)   /* bad argument                                   */
#define TOPO_AMD_EHIP (-2)     /* a HIP runtime call failed                      */
#define TOPO_AMD_ENODEV (-3)   /* no usable GPU / library not initialised        */
#define TOPO_AMD_ERCCL (-4)    /* an RCCL call failed                            */
#define TOPO_AMD_EUNSUP (-5)   /* valid request the kernels do not cover (yet)   */
#define TOPO_AMD_EEMPTY (-6)   /* Sx: a sector has no usable ray pixel; its plane was zero-filled and the
                                  other sectors were computed (nanmax over nothing, topo.py:947-951) */

/* descriptor ids for topo_amd_halo_rows */
#define TOPO_AMD_DESC_TPI 0
#define TOPO_AMD_DESC_STD 1
#define TOPO_AMD_DESC_GAUSS 2
#define TOPO_AMD_DESC_GRADIENT 3
#define TOPO_AMD_DESC_SOBEL 4
#define TOPO_AMD_DESC_SX 5
#define TOPO_AMD_DESC_VALLEY_RIDGE 6 /* p0 = side of the largest rotated kernel */

/* resolution layout for the gradient normalisation (reference topo.py:688-712) */
#define TOPO_AMD_RES_SCALAR 0 /* res_x[0], res_y[0]                                   */
#define TOPO_AMD_RES_1D 1     /* res_x[nx] per column, res_y[gny] per GLOBAL row      */
#define TOPO_AMD_RES_2D 2     /* res_x, res_y: [out_rows x nx] aligned with `out`     */

/* ---- runtime ------------------------------------------------------------------------ */
const char* topo_amd_version(void);
const char* topo_amd_last_error(void);
int topo_amd_device_count(void);
int topo_amd_init(int device);            /* idempotent for the same device           */
int topo_amd_shutdown(void);
int topo_amd_device_name(char* buf, int buflen);
/* Compute units the persistent kernels size their grids for: the device's CU count, or the
 * smaller TOPO_AMD_CU_LIMIT from the environment at init.  Results never depend on it.   */
int topo_amd_cu_count(void);

int topo_amd_malloc(void** dptr, size_t bytes);
int topo_amd_free(void* dptr);
/* Page-locked host memory for arrays handed to the host-buffer entry points (copies at the link's rate). */
int topo_amd_host_alloc(void** hptr, size_t bytes);
int topo_amd_host_free(void* hptr);
int topo_amd_memcpy_h2d(void* dst, const void* src, size_t bytes);
int topo_amd_memcpy_d2h(void* dst, const void* src, size_t bytes);
int topo_amd_memcpy_d2d(void* dst, const void* src, size_t bytes);
int topo_amd_memset(void* dst, int value, size_t bytes);
int topo_amd_sync(void);

/* ---- what kernel routing may know about a raster ------------------------------------------------------------------------
 * Results never depend on how a raster is cut into row blocks or shards, nor on what the library has seen before.  Nearly
 * all kernel choices cannot change a bit (integer sums are exact whoever forms them).  Two can, in the last bits: the
 * Gaussian / gradient of a raster whose ORDINARY values lie beyond 1e5 (a DEM in millimetres) runs on the vector-ALU kernels
 * instead of the f16 matrix-core ones, and TPI alone on fractional elevations (discs from 19 px) sums x in units of 2^-k m
 * with k = 8 ... 16 chosen from the raster's value range.  Both are properties of the WHOLE raster - its "class", read off a
 * lattice of about 16 K samples of the GLOBAL grid (rows / columns step / 2 + i step, step = extent / 128):
 *   - a block that is the whole raster (in_row0 == 0, in_rows == gny) is scanned by the library at the first call that
 *     needs the class and remembered with the buffer until the library writes or frees the buffer - or the caller, having
 *     written it with kernels of its own, says so with topo_amd_dem_changed;
 *   - the host-buffer entry points scan the caller's array;
 *   - a PARTIAL row block uses the class that was declared FOR ITS MEMORY: topo_amd_raster_scan_dev adds up the lattice
 *     points of the rows each block owns (counts += {samples, samples finite and beyond 1e5, samples with a fractional
 *     part}; range = {min, max} of the samples within +-2^18; start from 0, 0, 0 and +inf, -inf), then
 *     topo_amd_raster_class_from_scan(block, in_rows, gny, nx, counts, range) declares the sum for the device rows
 *     [block, block + in_rows * nx) of a gny x nx raster - once per block the application holds (the share of fractional
 *     samples only picks which exact disc kernels run first: time, never bits).  A later call whose `in` pointer lies
 *     inside declared rows of a raster of the same shape takes that class.  A declaration lives exactly as long as the
 *     data it describes: it is dropped when the library writes or frees memory overlapping the rows (uploads, copies,
 *     memset, an output plane, topo_amd_free), by topo_amd_dem_changed, or by topo_amd_raster_class_set(..., large = -1,
 *     ...) (block == NULL: every declaration).  It is not a property of a thread or of the process: two rasters held in
 *     one process never see each other's class.
 *   - the topo_amd_shard_* calls declare the class of their shard THEMSELVES at the first call on a buffer nothing is
 *     declared for (topo_amd_shard_classify: the scans of all ranks, added up by an all-reduce - collective like the
 *     calls themselves), so every shard takes the single GPU's kernels without the application doing anything.
 *   - a partial row block of the *_dev entry points that nothing was declared for is taken for an ordinary DEM in whole
 *     metres (large = 0, range 0 ... 4096, no fractional samples): a fixed rule, not a memory of earlier calls.  On such
 *     a raster the blocks give the whole raster's bits; on a raster of another class (fractional elevations within a
 *     few hundred metres, millimetres) they stay inside the tolerances below but may differ from the whole raster in
 *     the last bits of TPI / the Gaussian - declare the class to get its bits.                                           */
int topo_amd_dem_changed(const void* dptr, size_t bytes); /* bytes == 0: the whole allocation dptr lies in */
int topo_amd_raster_scan_dev(const float* in, int in_rows, int in_row0, int gny, int nx, int own_row0, int own_rows,
                             uint64_t counts[3], float range[2]);
int topo_amd_raster_class_from_scan(const float* block, int in_rows, int gny, int nx, const uint64_t counts[3],
                                    const float range[2]);
int topo_amd_raster_class_set(const float* block, int in_rows, int gny, int nx, int large, float lo, float hi,
                              float frac_share);
/* The class a call on a block starting at `block` (any row of declared rows) of a gny x nx raster would take; *declared = 0:
 * nothing is declared there (the values are then the ordinary DEM's).                                                     */
int topo_amd_raster_class_get(const float* block, int gny, int nx, int* declared, int* large, float* lo, float* hi,
                              float* frac_share);

/* HIP-event stopwatch on the compute stream (what bench.py times kernels with). */
int topo_amd_timer_start(void);
int topo_amd_timer_stop(float* elapsed_ms); /* records, synchronises, returns ms        */
/* Numbered HIP events on the compute stream (index 0 .. 511): one per launch boundary gives the
 * duration of every launch of a timed loop without a host synchronise inside the loop (bench.py
 * reports the median next to the mean).  topo_amd_mark_elapsed waits for mark `to`.         */
int topo_amd_mark(int index);
int topo_amd_mark_elapsed(int from, int to, float* elapsed_ms);

/* Deterministic synthetic terrain (float32 metres) written on the device: the value depends
 * only on (global row, column, seed), so shards agree on overlaps.  integer_valued != 0 rounds
 * to whole metres (SRTM/DHM25-like; TPI then takes the one-pass path), 0 keeps the fractional
 * part (swissALTI3D-like; every tile takes the two-pass path).                          */
int topo_amd_synth_dem_dev(float* out, int rows, int row0, int nx, uint32_t seed, int integer_valued);

/* ---- geometry helpers (host) -------------------------------------------------------- */
/* Tap count of the reference's circular_kernel(size) (topo.py:191-213).               */
int topo_amd_disc_tap_count(int size);
/* Writes the size x size 0/1 mask (row-major float32).                                 */
int topo_amd_disc_mask(int size, float* mask);
/* Ghost rows a row block needs above / below its output rows for one descriptor.
 * p0: size (TPI/STD) | sigma (GAUSS: of axis 0; GRADIENT: the `sigma` argument of the gradient)
 * p1: pre-smoothing sigma (TPI/STD, 0 = none) | sig_ratio (GRADIENT; 0 or 1 = isotropic) | unused
 * GRADIENT answers exactly the depth topo_amd_shard_gradient(sigma, sig_ratio) lays its block out with.
 * For SX pass the extremes of the offset table instead: p0 = -min(dj), p1 = max(dj).
 * For Gaussian radii int(4 sigma + 0.5) of 4 ... 15 (the gradient's too; also the pre-smoothing of TPI / STD) the
 * answer is 16 rows (gradient: 17) rather than the radius: the matrix-core kernels used there take a per-tile
 * offset sample 16 rows into their 32-row tiles.  A block with fewer ghost rows (but at least the radius) is still
 * computed correctly, by the vector-ALU kernels; only bit-identity with other partitions is lost.              */
int topo_amd_halo_rows(int descriptor, double p0, double p1, int* above, int* below);

/* ---- descriptors, device row-block form --------------------------------------------- */
/* TPI and/or STD over the reference's disc (replaces topo.tpi topo.py:144-181 and topo.std
 * topo.py:272-307, called from topo.py:138 and :266).  Either output may be NULL.  STD is
 * float32 on the device (the Python wrapper widens to float64 like the reference).       */
/* Arithmetic.  Sums of trunc(x), trunc(x)^2 and of the fractional parts (units of 2^-16 m) are exact integers in every
 * kernel, so the values do not depend on tiles, kernels or row blocks.  A sample that is not finite or beyond +-2^24 is
 * MISSING: exactly the pixels whose disc holds one are NaN.  Samples beyond +-2^18 (a raster in millimetres) and windows
 * with more relief than the 32-bit chains hold (nodata like -9999 next to terrain) take exact multi-limb passes (slower).
 * TPI alone (std_out == NULL), discs of 19 ... 101 px, windows with fractional elevations: the neighbourhood sum is taken
 * on x in units of 2^-k m (one integer chain; csrc/disc_wave_impl.hpp, tpi_scaled_march_kernel), k = 8 for an ordinary DEM
 * (at most 2^-9 m = 1.95 mm off per sample and therefore on TPI), up to 16 for rasters of small values (the raster class,
 * above); an output row whose own window holds more relief than that chain unwraps exactly is computed by the exact
 * general kernel - decided per row on the rows every row block that computes it holds.  Whole-metre DEMs, smaller discs
 * and the fused TPI + STD call are exact to float32 rounding.  TOPO_AMD_TPI_FRACTION_EXACT=1: the exact two-pass route.  */
int topo_amd_tpi_std_dev(const float* in, int in_rows, int in_row0, int gny, int nx, int size,
                         int out_row0, int out_rows, float* tpi_out, float* std_out);

/* TPI for several disc sizes of one resident block (the scale loop of compute_tpi, reference
 * topo.py:132-141, and of scripts/compute_topo_descriptors.py:25-38): sizes that have a two-disc
 * kernel (pairs out of 5, 7, 9, 11 px) are evaluated two at a time from ONE pass over the DEM,
 * the rest one by one; every plane has the bits of topo_amd_tpi_std_dev for its size.
 * tpi_outs[k] receives the plane of sizes[k] (out_rows x nx each).                         */
int topo_amd_tpi_multi_dev(const float* in, int in_rows, int in_row0, int gny, int nx, int n_sizes,
                           const int32_t* sizes, int out_row0, int out_rows, float* const* tpi_outs);
/* ndimage.gaussian_filter(dem, (sigma_y, sigma_x)), reflect boundary, truncate 4 sigma
 * (replaces topo.dem topo.py:62-80 and the pre-smoothing at topo.py:173, :298).  A sigma of
 * 0 skips that axis.  The intermediate plane lives in the library's own workspace.       */
/* Gaussian / gradient, matrix-core routes (filter radius 4 ... 121): a sample that is not finite, or larger than 1e5 in
 * magnitude (a sentinel like 1e20), is "wild": its outputs are recomputed by slower repair passes with exactly
 * ndimage.gaussian_filter's footprint (a block on which the fused short-filter kernel met one is remembered and takes the
 * two-pass kernels on later calls: the same bits).  A raster whose ORDINARY values lie beyond 1e5 (a DEM in millimetres)
 * runs on the vector-ALU kernels, which have no such limit (8192^2, sigma 3.25 / 13: 0.45 / 0.71 ms against 21.9 / 68.3 ms
 * through the repair passes): a property of the whole raster (the raster class, above), so every row block of it takes
 * the same kernels.  TOPO_AMD_GAUSS_LARGE_SAMPLE=0: always the matrix-core kernels.                                    */
int topo_amd_gaussian_dev(const float* in, int in_rows, int in_row0, int gny, int nx,
                          double sigma_y, double sigma_x, int out_row0, int out_rows,
                          float* out);

/* 3x3 Sobel pair / 8, true convolution, reflect (replaces topo.sobel topo.py:658-685).  */
int topo_amd_sobel_dev(const float* in, int in_rows, int in_row0, int gny, int nx,
                       int out_row0, int out_rows, float* dx_out, float* dy_out);

/* [dx, dy, slope, aspect] (replaces topo.gradient topo.py:597-644, called from :583).
 * sigma <= 1 -> Sobel; sig_ratio == 1 -> one isotropic smooth; otherwise the two
 * anisotropic smooths.  res_x / res_y are HOST double arrays for RES_SCALAR / RES_1D and
 * DEVICE float arrays for RES_2D.  Any output may be NULL.                                */
int topo_amd_gradient_dev(const float* in, int in_rows, int in_row0, int gny, int nx,
                          double sigma, double sig_ratio, int res_mode, const void* res_x,
                          const void* res_y, int out_row0, int out_rows, float* dx_out,
                          float* dy_out, float* slope_out, float* aspect_out);

/* Sx max-elevation-angle scan (replaces topo._sx_rolling topo.py:928-953, called from :856).
 * dj/di: offsets of the ray pixels relative to the target (host int32, n_off entries,
 * duplicates allowed); dist: metric distance of each (host double, NaN = skip, as left by
 * radius_min at topo.py:845); window: zero-frame width int(W/2); height: topo.py:947.     */
int topo_amd_sx_dev(const float* in, int in_rows, int in_row0, int gny, int nx,
                    const int32_t* dj, const int32_t* di, const double* dist, int n_off,
                    int window, double height, int out_row0, int out_rows, float* out);

/* Sx for n_az azimuth sectors in one pass over the DEM (the reference scans one
 * azimuth per call, topo.py:715-772 / :856, and its users loop).  Sector a owns entries
 * first[a] .. first[a+1]-1 of dj / di / dist (first: host int32, n_az + 1 entries), has the zero
 * frame window[a] and writes the device plane outs[a] (outs: host array of device pointers).
 * Neighbouring sectors overlap: ray pixels shared by several sectors are scanned once.  Every
 * plane has exactly the bits topo_amd_sx_dev gives for that sector alone.  A sector without a
 * usable ray pixel is treated as by topo_amd_sx_dev (plane zero-filled, TOPO_AMD_EEMPTY), but
 * only after the other sectors are done.                                                     */
int topo_amd_sx_multi_dev(const float* in, int in_rows, int in_row0, int gny, int nx, int n_az,
                          const int32_t* first, const int32_t* dj, const int32_t* di,
                          const double* dist, const int32_t* window, double height,
                          int out_row0, int out_rows, float* const* outs);

/* Valley / ridge index (replaces the angle loop of topo.valley_ridge, topo.py:431-447).
 * The host builds the kernels exactly as the reference does (V / U profiles topo.py:456-492,
 * quadratic-spline rotation and re-normalisation :515-525) and hands over, for each of
 * n_angles angles, n_planes (1..4) 2-D kernels of side ksize[a]: the sums of neighbouring
 * kernel planes that the reference's 3-D "same" convolution of the broadcast DEM applies
 * (DESIGN.md), flipped in both axes so that the device evaluates a correlation.  taps: HOST
 * float32, angle after angle, ksize[a]^2 taps in row-major order, 4 floats per tap (one per
 * plane, unused ones 0).  angles: HOST float32, the value stored in dir_out for each angle.
 * mean / stdev: the DEM is normalised as (x - mean) / stdev in float32 while it is read
 * (topo.py:427); topo_amd_mean_std_dev computes them for a device-resident DEM.
 * norm_out = max over angles and planes, clipped at 0; dir_out = first angle reaching it.
 * Three evaluations, chosen by the kernel tables alone (never by the data or the block):
 * rotated kernels of up to 25 cells a side run as a dense product on the matrix pipe
 * (split-f16 operands, float32 accumulation: closer to float64 than the float32 chain;
 * TOPO_AMD_VALLEY_MFMA_MAX_KERNEL, 0 = never) - over PAIRS of opposite cells when every
 * table is point-symmetric bit by bit, as the reference's are (rotated kernels of up to 120
 * cells a side: kernels of up to ~85 px; TOPO_AMD_VALLEY_FOLD=0: never), over the cells
 * otherwise (at most 240 cells with taps: up to 13 px); what neither takes tap by tap in
 * float32; of that, rotated kernels of 64 cells a side and more
 * (TOPO_AMD_VALLEY_FFT_MIN_KERNEL), and any too large for that kernel, by FFT like the
 * reference's signal.convolve.  The first two: a pixel whose kernel footprint holds a
 * non-finite sample is evaluated tap by tap in both, row blocks give the single block's bits.
 * FFT: same contract (1e-4 of the range), but a NaN in the block reaches every output and
 * row blocks agree to rounding instead of bit for bit.                                      */
int topo_amd_valley_ridge_dev(const float* in, int in_rows, int in_row0, int gny, int nx,
                              const float* taps, const int32_t* ksize, const float* angles,
                              int n_angles, int n_planes, double mean, double stdev,
                              int out_row0, int out_rows, float* norm_out, float* dir_out);
/* Which evaluation the calling thread's last valley / ridge call took (for tests and diagnostics): 0 tap by tap,
 * 1 matrix pipe, 2 FFT; + 4: the matrix-pipe pass was followed by the tap-by-tap kernel over the tiles in which it met
 * non-finite samples (launched whenever the matrix pipe is used; it returns at once in tiles that are not flagged);
 * + 8: the matrix pipe ran over pairs of opposite cells (point-symmetric tables); + 16: with the pixel operands streamed
 * in chunks (more than 240 pairs: kernels of 19 px and more).                                                            */
int topo_amd_valley_route(int* route);
/* Mean and population standard deviation (numpy's default ddof = 0) of count device floats,
 * accumulated in float64.                                                                */
int topo_amd_mean_std_dev(const float* in, size_t count, double* mean, double* stdev);

/* ---- descriptors, host-buffer form (single block, whole DEM) -------------------------- */
/* Upload, kernels and download run in row chunks on three streams (upload || kernels || download: row blocks give the
 * single block's bits), page-locked arrays (topo_amd_host_alloc) and pageable ones alike.  The device planes a call needs
 * are kept for the next call (grow-only); topo_amd_release_host_planes() gives them back to the device.                  */
int topo_amd_release_host_planes(void);
/* Row chunks the calling thread's last host-buffer call ran in (1: upload, kernels, download one after the other - arrays
 * of fewer than three chunks, or TOPO_AMD_HOST_PIPELINE=0; a multi-scale call: the most any scale ran in).  For tests and
 * diagnostics: which branch of the pipeline a call took.  Environment, read at every call: TOPO_AMD_HOST_CHUNK_MB (64; a
 * chunk is whole multiples of 960 rows, at least 960; the last one takes the rest: half a chunk to a chunk and a half),
 * TOPO_AMD_HOST_PIPELINE=0 (off), TOPO_AMD_HOST_DOWNLOADS=thread|inline
 * (who issues the downloads; default: the calling thread when every array is page-locked, a second thread otherwise).
 * topo_amd_valley_ridge_f32 is not pipelined (always one chunk).                                                         */
int topo_amd_host_chunks(int* chunks);
int topo_amd_tpi_f32(const float* dem, int ny, int nx, int size, double sigma, float* out);
int topo_amd_std_f32(const float* dem, int ny, int nx, int size, double sigma, float* out);
int topo_amd_tpi_std_f32(const float* dem, int ny, int nx, int size, double sigma,
                         float* tpi_out, float* std_out);
/* Several scales of TPI and / or STD from ONE upload of the DEM (the loop over scales of the reference's
 * compute_tpi / compute_std, topo.py:88-141 and :216-269, which calls tpi / std once per scale): sizes[k],
 * sigmas[k] (0 = no pre-smoothing) -> tpi_outs[k], std_outs[k], each a host plane [ny x nx] or NULL (either
 * array of planes may itself be NULL).  Plane k has the bits of topo_amd_tpi_std_f32(sizes[k], sigmas[k]);
 * the call moves (1 + planes) x ny x nx x 4 bytes over PCIe instead of (scales + planes) x that.          */
int topo_amd_tpi_std_multi_f32(const float* dem, int ny, int nx, int n_scales, const int32_t* sizes,
                               const double* sigmas, float* const* tpi_outs, float* const* std_outs);
int topo_amd_gauss_f32(const float* dem, int ny, int nx, double sigma_y, double sigma_x,
                       float* out);
int topo_amd_sobel_f32(const float* dem, int ny, int nx, float* dx_out, float* dy_out);
/* res_mode RES_2D here takes HOST float arrays [ny x nx].                                 */
int topo_amd_gradient_f32(const float* dem, int ny, int nx, double sigma, double sig_ratio,
                          int res_mode, const void* res_x, const void* res_y, float* dx_out,
                          float* dy_out, float* slope_out, float* aspect_out);
int topo_amd_sx_f32(const float* dem, int ny, int nx, const int32_t* dj, const int32_t* di,
                    const double* dist, int n_off, int window, double height, float* out);
/* outs: host array of n_az HOST planes [ny x nx].                                          */
int topo_amd_sx_multi_f32(const float* dem, int ny, int nx, int n_az, const int32_t* first,
                          const int32_t* dj, const int32_t* di, const double* dist,
                          const int32_t* window, double height, float* const* outs);
int topo_amd_valley_ridge_f32(const float* dem, int ny, int nx, const float* taps,
                              const int32_t* ksize, const float* angles, int n_angles,
                              int n_planes, double mean, double stdev, float* norm_out,
                              float* dir_out);

/* ---- row sharding over the GPUs of one node (RCCL over xGMI) -------------------------- */
/* The reference's only precedent is dask map_overlap(depth, boundary="none") for TPI
 * (topo.py:177-178): independent blocks plus ghost rows.  Rank r owns a contiguous row
 * block; ghost rows travel with one ncclSend/ncclRecv pair per neighbour.                */
#define TOPO_AMD_UNIQUE_ID_BYTES 128
int topo_amd_comm_unique_id(char id[TOPO_AMD_UNIQUE_ID_BYTES]); /* rank 0, then broadcast */
int topo_amd_comm_init(int rank, int nranks, const char id[TOPO_AMD_UNIQUE_ID_BYTES]);
int topo_amd_comm_rank(void);
int topo_amd_comm_size(void);
int topo_amd_comm_destroy(void);

/* `block` holds [halo_above | rows_local | halo_below] rows of nx floats.  Fills the ghost
 * rows from the neighbours on the communication stream (ranks at the global edge skip the
 * missing side) and records an event; topo_amd_halo_wait() makes the compute stream wait
 * for it.  Every rank must call with the same halo depths.  Requires rows_local >= both.  */
/* Loop-back mode (environment TOPO_AMD_HALO_LOOPBACK=1, communicator of ONE rank): the same
 * ncclSend / ncclRecv pairs are issued to rank 0 itself, with periodic wrap - the block's last
 * halo_above rows land in its top ghost rows, its first halo_below rows in its bottom ghost
 * rows - so that pointer offsets, counts, stream ordering and the CU reservation of the
 * exchange run on a single GPU (tests/test_gpu_halo_loopback.py).                          */
int topo_amd_halo_exchange_start(float* block, int rows_local, int nx, int halo_above,
                                 int halo_below);
int topo_amd_halo_wait(void);
/* Collective: the class of the WHOLE sharded raster (above), declared for this rank's owned rows.  owned: the first row this
 * rank owns (device pointer), rows_local of them starting at global row row0.  The topo_amd_shard_* calls run it themselves
 * when nothing is declared for the shard (first call, or after the shard's rows were rewritten through the library); calling
 * it is only needed after the application rewrote the rows with kernels of its own (or use topo_amd_dem_changed).  Refused
 * when the shard is part of a raster and no communicator exists.  In loop-back mode the rank's rows are scanned at every
 * placement of the periodic stack they stand for.                                                                        */
int topo_amd_shard_classify(const float* owned, int rows_local, int row0, int gny, int nx);
/* Round 4: a sharded call is ONE launch per kernel - the interior rows, then, behind a device-side gate the
 * communication stream opens when the ghost rows have landed, the seam rows - with a clean-up launch behind the
 * exchange's event for blocks that found the gate closed for longer than TOPO_AMD_GATE_WAIT_US (100).  Reads and
 * resets the number of such blocks since the last call: 0 when the exchange hid behind the interior rows.      */
int topo_amd_gate_giveups(unsigned* count);
/* Declares the ghost depth the shard buffers handed to topo_amd_shard_* are laid out with:
 * [halo_above | rows_local | halo_below] rows.  A descriptor that needs fewer ghost rows uses
 * the ones next to the owned rows; one that needs more is refused (TOPO_AMD_EINVAL) instead of
 * reading the owned rows from the wrong offset and receiving past the end of the buffer.
 * -1 / -1 (the default): the buffer has exactly the depth topo_amd_halo_rows gives for the
 * descriptor of each call.  The declaration belongs to the CALLING THREAD (thread-local), so
 * concurrent drivers of differently laid-out shards do not disturb each other; a thread that
 * has never declared one uses the last layout any thread declared (set up once, drive from
 * worker threads).  topo_amd_shard_layout_get reads back what applies to the calling thread
 * (to save and restore around a call).                                                      */
int topo_amd_shard_layout(int halo_above, int halo_below);
int topo_amd_shard_layout_get(int* halo_above, int* halo_below);

/* Sharded TPI/STD step: ghost exchange overlapped with the interior rows, then the two
 * seam strips.  `block` as above with halo_above == halo_below == topo_amd_halo_rows(TPI).
 * row0 = global index of the first local row.  Outputs are rows_local x nx.               */
int topo_amd_shard_tpi_std(float* block, int rows_local, int row0, int gny, int nx, int size,
                           float* tpi_out, float* std_out);
int topo_amd_shard_gradient(float* block, int rows_local, int row0, int gny, int nx,
                            double sigma, double sig_ratio, int res_mode, const void* res_x,
                            const void* res_y, float* dx_out, float* dy_out, float* slope_out,
                            float* aspect_out);
int topo_amd_shard_sx(float* block, int rows_local, int row0, int gny, int nx,
                      const int32_t* dj, const int32_t* di, const double* dist, int n_off,
                      int window, double height, float* out);
/* halo_above / halo_below: the largest -dj / dj over the usable ray pixels of all sectors;
 * one ghost-row exchange serves every sector.                                               */
int topo_amd_shard_sx_multi(float* block, int rows_local, int row0, int gny, int nx, int n_az,
                            const int32_t* first, const int32_t* dj, const int32_t* di,
                            const double* dist, const int32_t* window, double height,
                            float* const* outs);
/* Sharded valley / ridge index: `block` as above with halo_above / halo_below =
 * topo_amd_halo_rows(VALLEY_RIDGE, largest kernel side).  The mean and standard deviation of
 * the whole DEM come from float64 moments of the owned rows and one ncclAllReduce (the only
 * true collective on the path; exact, hence the same for every sharding, on a DEM of whole
 * metres).                                                                                  */
int topo_amd_shard_valley_ridge(float* block, int rows_local, int row0, int gny, int nx,
                                const float* taps, const int32_t* ksize, const float* angles,
                                int n_angles, int n_planes, float* norm_out, float* dir_out);

#ifdef __cplusplus
}
#endif
#endif /* TOPO_AMD_H */
