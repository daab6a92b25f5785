/*
 * vwgpu.h -- C ABI of the MI355X (gfx950) MODWT / SWT engine.
 *
 * Drop-in boundary for the batched MODWT path of Prophetizo/VectorWave.  The
 * reference has no FFI: its seam is a set of static Java methods that process
 * one signal with one filter (WaveletOperations.java:25-49).  That is too fine
 * a grain for a GPU, so the boundary moves up to whole-batch calls; every entry
 * point below names the reference method(s) it replaces (paths relative to
 * src/main/java/ai/prophetizo/wavelet/).  A Java 23 host binds these symbols
 * with java.lang.foreign downcalls over off-heap MemorySegments.
 *
 * Conventions
 *   - plain pointers and sizes only; every function returns a vw_status.
 *   - all data is IEEE-754 binary64; element counts are int64_t.
 *   - signals:       row-major [B][N]
 *   - coefficients:  row-major [B][J+1][N], bands ordered d_1, ..., d_J, a_J
 *                    (level 1 = finest; the single-level transform is J = 1,
 *                    i.e. [B][2][N] = detail, approximation).
 *   - filters are the wavelet's UNSCALED taps (what lowPassDecomposition() etc.
 *     return); the library applies the reference's `tap * (1.0/Math.sqrt(2.0))`
 *     scaling (modwt/MODWTTransform.java:133-142) itself, in binary64, so host
 *     and device agree bit for bit.
 *   - "_dev" entry points take DEVICE pointers (hipMalloc'd or any framework's
 *     device allocation on the context's GPU) and enqueue on `stream` (a
 *     hipStream_t passed as void*; NULL is HIP's default stream, as everywhere in
 *     HIP) without synchronising.  Multi-pass plans keep their intermediate
 *     plane in a buffer owned by the context, so calls that share a context must
 *     be ordered on one stream (use one context per concurrent stream).  The
 *     other entry points take HOST pointers, stage through device buffers owned
 *     by the context on the context's own stream, and return when the result
 *     is in the caller's buffer.  The caller owns every buffer it passes; the
 *     library keeps no pointer after return.
 *   - arithmetic is the reference's scalar path: products and sums rounded
 *     separately (no FMA contraction), taps added in ascending order.
 */
#ifndef VWGPU_H
#define VWGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vw_ctx vw_ctx;

/* Status codes.  1..7 and 103/104 carry the number of the reference's ErrorCode
 * (exception/ErrorCode.java:24-75: VAL_001..VAL_007, CFG_003, CFG_004). */
typedef enum vw_status {
    VW_OK = 0,
    VW_E_NULL = 1,            /* VAL_NULL_ARGUMENT: a required pointer is NULL         */
    VW_E_NONFINITE = 3,       /* VAL_NON_FINITE_VALUES: NaN/Inf in the input           */
    VW_E_NOT_POSITIVE = 4,    /* VAL_NOT_POSITIVE: negative threshold / count          */
    VW_E_TOO_LARGE = 5,       /* VAL_TOO_LARGE: level - 1 >= 31 or size overflow       */
    VW_E_EMPTY = 6,           /* VAL_EMPTY: N == 0                                      */
    VW_E_LENGTH_MISMATCH = 7, /* VAL_LENGTH_MISMATCH: filter lengths unusable           */
    VW_E_BOUNDARY = 103,      /* CFG_UNSUPPORTED_BOUNDARY_MODE                          */
    VW_E_LEVEL = 104,         /* CFG_INVALID_DECOMPOSITION_LEVEL                        */
    VW_E_UNSUPPORTED = 105,   /* CFG_UNSUPPORTED_OPERATION: e.g. FIXED without a value  */
    VW_E_NO_DEVICE = 900,     /* no usable gfx950 device / HIP runtime failure          */
    VW_E_HIP = 901,           /* a HIP call failed; see vw_last_error                   */
    VW_E_NOMEM = 902,         /* device or host allocation failed                       */
    VW_E_WORKSPACE = 903      /* caller-supplied workspace too small                    */
} vw_status;

/* Boundary modes accepted by the MODWT classes (modwt/MODWTTransform.java:88-90). */
typedef enum vw_boundary {
    VW_BOUNDARY_PERIODIC = 0,
    VW_BOUNDARY_ZERO_PADDING = 1,
    VW_BOUNDARY_SYMMETRIC = 2
} vw_boundary;

/* Flags (bitwise OR). */
enum {
    /* Scan the input for NaN/Inf while it is loaded and report the first bad
     * flat index, as ValidationUtils.validateFiniteValues does for forward()/
     * decompose() (util/ValidationUtils.java:106-117).  Without it the call
     * behaves like forwardBatchOptimized, which skips validation
     * (modwt/MODWTTransform.java:537-587). */
    VW_FLAG_CHECK_FINITE = 1,
    /* SYMMETRIC single-level inverse only: reflect t+l as inverseBatchOptimized
     * does (modwt/MODWTTransform.java:644-657) instead of t-l as inverse() does
     * (:269-287).  The two reference methods disagree; the caller picks. */
    VW_FLAG_SYMMETRIC_BATCH_VARIANT = 2,
    /* Skip the reference's level-count check (1 <= J <= vw_max_levels). */
    VW_FLAG_NO_LEVEL_CHECK = 4,
    /* The call comes from the single-level class MODWTTransform (requires
     * levels == 1): any N >= 1 is accepted, filters are never truncated, and the
     * SYMMETRIC inverse adds each tap's two products as a pair
     * (modwt/MODWTTransform.java:269-287).  Without it the call follows
     * MultiLevelMODWTTransform: level-count check, filters longer than the
     * signal truncated to its length (modwt/MultiLevelMODWTTransform.java:
     * 801-806, 628-636), SYMMETRIC inverse adding the low-pass and high-pass
     * products to the sum one after the other (:664-685). */
    VW_FLAG_SINGLE_LEVEL = 8,
    /* Opt out of bit parity: let the tiled kernels fuse each multiply-add (one rounding per tap
     * instead of two).  Half the fp64 instructions; every coefficient stays within
     * 1e-12 * max(|ref|, max|x|) of the default arithmetic (measured ~1e-15), but is no longer
     * bit-identical to the reference's scalar Java path.  Ignored by the one-level generic
     * kernels (SYMMETRIC, truncated filters). */
    VW_FLAG_FMA = 16
};

/* ---- context ----------------------------------------------------------- */

/* Create a context on HIP device `device` (one context per GPU; the multi-GPU
 * layout is one process, or one context, per device with rows sharded by the
 * host).  Fails with VW_E_NO_DEVICE if the device is missing. */
int32_t vw_init(int32_t device, vw_ctx **out_ctx);
int32_t vw_destroy(vw_ctx *ctx);
/* Message of the last failure on this context (never NULL; valid until the next call). */
const char *vw_last_error(const vw_ctx *ctx);
/* Human-readable description of the execution back end; the truthful
 * counterpart of WaveletOperations.getPerformanceInfo (WaveletOperations.java:56-72). */
const char *vw_performance_info(const vw_ctx *ctx);
/* Block until everything enqueued on the context's own stream has finished. */
int32_t vw_synchronize(vw_ctx *ctx);
/* Number of HIP devices visible to the process (0 if none / no runtime). */
int32_t vw_device_count(void);

/* Choose among equivalent launch plans of this context (no counterpart in the reference; for tests and
 * tuning).  Every setting yields the same results bit for bit -- the options only select tile shapes, run
 * lengths and kernel families that the planner would otherwise pick by size, so that small test inputs can
 * reach the plans large inputs take.  The library never reads environment variables.  Names: "fwd_tile" /
 * "inv_tile" (value = chunks per tile * 100 + halo chunks, e.g. 401; 0 = planner's choice), "fwd_chunks"
 * (tiles per run of the streaming forward), "stream_wgs" (workgroups wanted per streaming launch),
 * "sdry_chunks" (run length of the long-context kernels), "no_sdry", "no_unroll", "no_reg_select",
 * "no_wg_select" (0/1: rows of 2049..32768 samples through the 256-lane radix select),
 * "select_lds_cap" (longest row the 256-lane selection kernel copies into LDS; 0 = all that fits),
 * "no_std_chains", "no_wide_inv", "no_vec2" (0/1), "no_ring" / "no_blk" (0/1: inverse tile kernel without the
 * LDS-DMA band ring / without register-blocked windows), "carry_chunks" (>= 2: the ring kernel over runs of that
 * many tiles with the right context carried from tile to tile; 0 = one tile per workgroup, the default),
 * "no_row_denoise" / "no_band_regs" (0/1: short-row
 * denoise through the three-launch path / with its bands in LDS instead of registers), "inv_vm" (0, 1, 2: one-
 * sample, 8-byte-aligned pair, 16-byte pair accesses of the long-context inverse), "nt_stores" (-1 by launch
 * size, 0, 1), "inv_wave" (one wavefront per run of 512-sample chunks for the six-level inverse: 0 = where it pays,
 * -1 = never, N = wherever it applies with runs of N chunks) / "inv_wave_nt" (-1, 0, 1: its band loads carry the
 * streaming hint), "no_sym_tiles" (0/1: SYMMETRIC through the one-level kernels instead of the tile kernels),
 * "no_soa_runs" (0/1: the structure-of-arrays transform one output per lane) / "soa_run_len" (8, 16, 32: samples per
 * lane of its register-window kernel; 0 = default), "reset" (all defaults).  Unknown names return VW_E_UNSUPPORTED. */
int32_t vw_set_plan_option(vw_ctx *ctx, const char *name, int64_t value);
/* The kernels that the context's last forward and last inverse call enqueued, as text
 * ("forward: k_fwd_stream[1-6]; inverse: k_inv_ring[1-6]": kernel family and the levels each launch covers; a
 * single-launch denoise names its kernel on both sides).  For benchmarks and profiles that must name the kernel
 * they measured; no counterpart in the reference.  Valid until the next call on the context. */
const char *vw_last_plan(vw_ctx *ctx);

/* Page-locked host memory for the HOST-pointer entry points.  They accept any host memory, but copies to and
 * from pageable memory go through the runtime's bounce buffers at a fraction of the PCIe rate; buffers obtained
 * here are copied by DMA directly.  (No counterpart in the reference: a Java caller wraps the address in a
 * MemorySegment of `bytes` bytes instead of allocating from an Arena.) */
int32_t vw_host_alloc(vw_ctx *ctx, size_t bytes, void **out);
int32_t vw_host_free(vw_ctx *ctx, void *ptr);

/* ---- pure host helpers ------------------------------------------------- */

/* MultiLevelMODWTTransform.calculateMaxLevels / getMaximumLevels
 * (modwt/MultiLevelMODWTTransform.java:477-523, :776-778): largest j < 10 with
 * (L-1)*2^(j-1)+1 <= N; 0 when N <= L. */
int32_t vw_max_levels(int64_t n, int32_t filter_len);

/* ---- forward ------------------------------------------------------------ */

/* Batched J-level forward MODWT.  Replaces MODWTTransform.forward /
 * forwardBatch / forwardBatchOptimized for J = 1 (modwt/MODWTTransform.java:
 * 123-179, 459-487, 537-587) and MultiLevelMODWTTransform.decompose /
 * decomposeMutable + transformAtLevel + applyScaledMODWT for J >= 1
 * (modwt/MultiLevelMODWTTransform.java:214-353, 528-538, 793-834), which in turn
 * replace WaveletOperations.circularConvolveMODWT / zeroPaddingConvolveMODWT /
 * symmetricConvolveMODWT (WaveletOperations.java:25-49).
 *   x [B][N] in, coeffs [B][J+1][N] out; h/g = lowPass/highPassDecomposition().
 *   Enforces 1 <= J <= vw_max_levels(N, lh) like decompose() unless
 *   VW_FLAG_SINGLE_LEVEL or VW_FLAG_NO_LEVEL_CHECK is given.
 *   first_bad (may be NULL): with VW_FLAG_CHECK_FINITE receives the flat index
 *   b*N+t of the first non-finite input, or -1.  The host variant then returns
 *   VW_E_NONFINITE; the _dev variant writes the int64 through the DEVICE
 *   pointer first_bad_dev asynchronously and always returns VW_OK. */
int32_t vw_modwt_forward_batch(vw_ctx *ctx, const double *x, int64_t b, int64_t n, int32_t levels,
                               const double *h, int32_t lh, const double *g, int32_t lg,
                               int32_t boundary, uint32_t flags, double *coeffs,
                               int64_t *first_bad);
int32_t vw_modwt_forward_batch_dev(vw_ctx *ctx, const double *x_dev, int64_t b, int64_t n,
                                   int32_t levels, const double *h, int32_t lh, const double *g,
                                   int32_t lg, int32_t boundary, uint32_t flags,
                                   double *coeffs_dev, int64_t *first_bad_dev, void *stream);

/* ---- inverse ------------------------------------------------------------ */

/* Batched J-level inverse MODWT.  Replaces MODWTTransform.inverse /
 * inverseBatch / inverseBatchOptimized for J = 1 (modwt/MODWTTransform.java:
 * 193-291, 504-532, 592-662) and MultiLevelMODWTTransform.reconstruct +
 * reconstructSingleLevel + applyScaledInverseMODWT for J >= 1
 * (modwt/MultiLevelMODWTTransform.java:362-375, 543-553, 621-689).
 *   coeffs [B][J+1][N] in, x [B][N] out; hr/gr = lowPass/highPassReconstruction().
 *   PERIODIC / ZERO_PADDING loop l < lhr and index gr[l] like the reference, so
 *   lgr >= lhr is required (VW_E_LENGTH_MISMATCH otherwise -- the reference
 *   throws ArrayIndexOutOfBounds there); extra high-pass taps are ignored.
 *   VW_FLAG_CHECK_FINITE checks the coefficients (MODWTResult.isValid,
 *   modwt/MODWTTransform.java:196-207). */
int32_t vw_modwt_inverse_batch(vw_ctx *ctx, const double *coeffs, int64_t b, int64_t n,
                               int32_t levels, const double *hr, int32_t lhr, const double *gr,
                               int32_t lgr, int32_t boundary, uint32_t flags, double *x,
                               int64_t *first_bad);
int32_t vw_modwt_inverse_batch_dev(vw_ctx *ctx, const double *coeffs_dev, int64_t b, int64_t n,
                                   int32_t levels, const double *hr, int32_t lhr, const double *gr,
                                   int32_t lgr, int32_t boundary, uint32_t flags, double *x_dev,
                                   int64_t *first_bad_dev, void *stream);

/* Band-selective reconstruction: like vw_modwt_inverse_batch but band k of the
 * coefficient block is treated as all zeros unless bit k of `band_mask` is set
 * (bit 0 = d_1, ..., bit J-1 = d_J, bit J = a_J).  Replaces
 * MultiLevelMODWTTransform.reconstructFromLevel / reconstructLevels
 * (modwt/MultiLevelMODWTTransform.java:385-468) and
 * VectorWaveSwtAdapter.extractLevel (swt/VectorWaveSwtAdapter.java:326-348). */
int32_t vw_modwt_inverse_bands_batch(vw_ctx *ctx, const double *coeffs, int64_t b, int64_t n,
                                     int32_t levels, const double *hr, int32_t lhr,
                                     const double *gr, int32_t lgr, int32_t boundary,
                                     uint32_t band_mask, double *x);
int32_t vw_modwt_inverse_bands_batch_dev(vw_ctx *ctx, const double *coeffs_dev, int64_t b,
                                         int64_t n, int32_t levels, const double *hr, int32_t lhr,
                                         const double *gr, int32_t lgr, int32_t boundary,
                                         uint32_t band_mask, double *x_dev, void *stream);

/* ---- SWT denoise -------------------------------------------------------- */

/* Batched VectorWaveSwtAdapter.denoise(signal, levels, threshold, soft)
 * (swt/VectorWaveSwtAdapter.java:282-312): decompose, threshold the detail
 * bands 1..J (approximation untouched), reconstruct.
 *   threshold < 0 selects the universal threshold per signal
 *   (applyUniversalThreshold / estimateNoiseSigma, :255-270, :377-395):
 *   sigma = median(|d_1|) / 0.6745 with the exact order statistics the
 *   reference's sort yields (even N: mean of elements N/2-1 and N/2), thr =
 *   sigma * sqrt_2_ln_n.  The host passes sqrt_2_ln_n = Math.sqrt(2 *
 *   Math.log(N)) so that the transcendental is evaluated by the host's libm;
 *   pass a negative value to let the library compute it with its own libm.
 *   soft != 0: |c| > thr ? signum(c) * (|c| - thr) : 0.0; soft == 0: |c| <= thr
 *   -> 0.0 (modwt/MutableMultiLevelMODWTResult.java:97-114).
 *   x [B][N] in, y [B][N] out (may alias x).  thresholds_out (may be NULL)
 *   receives the B thresholds used.
 *   The _dev variant needs a scratch block for the coefficients: pass
 *   workspace_dev/workspace_bytes >= vw_swt_denoise_workspace_bytes(...), or
 *   NULL/0 to let the context allocate (and keep) one. */
int32_t vw_swt_denoise_batch(vw_ctx *ctx, const double *x, int64_t b, int64_t n, int32_t levels,
                             const double *h, int32_t lh, const double *g, int32_t lg,
                             const double *hr, int32_t lhr, const double *gr, int32_t lgr,
                             int32_t boundary, double threshold, double sqrt_2_ln_n, int32_t soft,
                             double *y, double *thresholds_out);
int32_t vw_swt_denoise_batch_dev(vw_ctx *ctx, const double *x_dev, int64_t b, int64_t n,
                                 int32_t levels, const double *h, int32_t lh, const double *g,
                                 int32_t lg, const double *hr, int32_t lhr, const double *gr,
                                 int32_t lgr, int32_t boundary, double threshold,
                                 double sqrt_2_ln_n, int32_t soft, double *y_dev,
                                 double *thresholds_out_dev, void *workspace_dev,
                                 size_t workspace_bytes, void *stream);
size_t vw_swt_denoise_workspace_bytes(int64_t b, int64_t n, int32_t levels);

/* vw_swt_denoise_batch_dev for callers that cannot allocate device memory (no counterpart in the reference):
 * code under HIP graph capture, tight serving loops over fixed-size blocks, harnesses that forbid hipMalloc.
 * Same arguments, kernels and results (bit for bit) as vw_swt_denoise_batch_dev without its workspace
 * parameters: the coefficient block, the per-row thresholds and the intermediate plane of multi-pass plans live
 * in a static array of VW_NOALLOC_SCRATCH_BYTES bytes that is part of the library's device code -- one copy per
 * GPU, held from library load until the process ends.  The call makes no hipMalloc/hipFree and no other
 * allocating call.  Limit: vw_swt_denoise_workspace_bytes(b, n, levels) rounded up to 256 (plus b*n*8 for the
 * rare multi-pass plans) must not exceed VW_NOALLOC_SCRATCH_BYTES, else VW_E_WORKSPACE and nothing is
 * enqueued; 1000 rows of 4096 samples at J = 4 need 163,848,192 bytes and fit, 1024 such rows do not.  (Rows
 * that the single-launch short-row kernels take -- up to 2048 samples with 2-, 4- or 8-tap filters, such as the
 * 8192 x 480, J = 4 streaming block -- never touch the array and are not limited by it.)  Every call on one GPU uses that
 * GPU's one array (whatever the context or stream) and leaves its last contents there: calls that overlap in
 * time on one GPU race and must be ordered by the caller (one stream, or events); nothing in the array needs
 * clearing between calls, every byte a call reads was written by that call.  Recording the call into a HIP
 * graph and replaying it on changed input is covered by tests/test_gpu_threads_capture.py.  Cost: the array
 * takes VW_NOALLOC_SCRATCH_BYTES of device memory on every GPU the process loads the library on, whether or
 * not this entry point is ever called; if its address cannot be looked up at vw_init the other entry points
 * work as usual and this one returns VW_E_WORKSPACE for the batches that would need the array. */
#define VW_NOALLOC_SCRATCH_BYTES ((size_t)160 << 20)
int32_t vw_swt_denoise_batch_noalloc_dev(vw_ctx *ctx, const double *x_dev, int64_t b, int64_t n,
                                         int32_t levels, const double *h, int32_t lh, const double *g,
                                         int32_t lg, const double *hr, int32_t lhr, const double *gr,
                                         int32_t lgr, int32_t boundary, double threshold,
                                         double sqrt_2_ln_n, int32_t soft, double *y_dev,
                                         double *thresholds_out_dev, void *stream);

/* ---- WaveletDenoiser --------------------------------------------------------- */

/* Threshold selection methods of denoising/WaveletDenoiser.java:588-617. */
typedef enum vw_threshold_method {
    VW_THRESHOLD_UNIVERSAL = 0, /* sigma * sqrt(2 ln n)                          (:389-391) */
    VW_THRESHOLD_SURE = 1,      /* min of Stein's risk over |c_k|, capped        (:432-483) */
    VW_THRESHOLD_MINIMAX = 2,   /* piecewise in n                                (:488-499) */
    VW_THRESHOLD_BAYES = 3,     /* sigma^2 / sqrt(max(var - sigma^2, 0) + 1e-10) (:512-539) */
    VW_THRESHOLD_FIXED = 4      /* caller's value; single-level form only        (:350-360) */
} vw_threshold_method;

/* Batched WaveletDenoiser (denoising/WaveletDenoiser.java).
 *   flags & VW_FLAG_SINGLE_LEVEL, levels == 1: denoise(signal, method, type) (:121-139), or
 *     denoiseFixed(signal, fixed_threshold, type) (:350-360) when method == VW_THRESHOLD_FIXED:
 *     MODWTTransform.forward, sigma = median(|d|)/0.6745, threshold the detail band,
 *     MODWTTransform.inverse.
 *   otherwise: denoiseMultiLevel(signal, levels, method, type) (:152-167, :179-227): sigma from
 *     level 1, per level threshold = calculateThreshold(d_j, sigma / Math.sqrt(1 << (j-1)), method);
 *     VW_THRESHOLD_FIXED is VW_E_UNSUPPORTED here, as calculateThreshold throws for it (:405-415).
 *   soft != 0: SOFT, else HARD (:544-568).  log_n = Math.log(N) from the host's libm (negative: the
 *   library computes it).  Mean, variance and SURE risks are summed in index order like the Java
 *   loops, so thresholds and outputs are bit-identical to the scalar reference.  SURE costs O(N^2)
 *   per band, as it does in the reference.
 *   x [B][N] in, y [B][N] out (may alias x); thresholds_out (may be NULL) receives [B][levels].
 *   The _dev variant takes a scratch block of vw_wavelet_denoise_workspace_bytes(...) or NULL/0 to let
 *   the context allocate (and keep) one. */
int32_t vw_wavelet_denoise_batch(vw_ctx *ctx, const double *x, int64_t b, int64_t n, int32_t levels,
                                 const double *h, int32_t lh, const double *g, int32_t lg,
                                 const double *hr, int32_t lhr, const double *gr, int32_t lgr,
                                 int32_t boundary, uint32_t flags, int32_t method, int32_t soft,
                                 double fixed_threshold, double log_n, double *y, double *thresholds_out);
int32_t vw_wavelet_denoise_batch_dev(vw_ctx *ctx, const double *x_dev, int64_t b, int64_t n,
                                     int32_t levels, const double *h, int32_t lh, const double *g,
                                     int32_t lg, const double *hr, int32_t lhr, const double *gr,
                                     int32_t lgr, int32_t boundary, uint32_t flags, int32_t method,
                                     int32_t soft, double fixed_threshold, double log_n, double *y_dev,
                                     double *thresholds_out_dev, void *workspace_dev,
                                     size_t workspace_bytes, void *stream);
size_t vw_wavelet_denoise_workspace_bytes(int64_t b, int64_t n, int32_t levels);

/* WaveletDenoiser.calculateThreshold over a coefficient block [B][levels+1][N] (:385-427): sigma per
 * row from band 0, thresholds [B][levels] out; level_scaled != 0 divides sigma by sqrt(1 << (j-1)). */
int32_t vw_denoiser_thresholds_batch(vw_ctx *ctx, const double *coeffs, int64_t b, int64_t n,
                                     int32_t levels, int32_t method, int32_t level_scaled, double log_n,
                                     double *thresholds);
int32_t vw_denoiser_thresholds_batch_dev(vw_ctx *ctx, const double *coeffs_dev, int64_t b, int64_t n,
                                         int32_t levels, int32_t method, int32_t level_scaled,
                                         double log_n, double *thresholds_dev, double *sigma_scratch_dev,
                                         void *stream);

/* ---- streaming denoiser ---------------------------------------------------- */

/* MODWTStreamingDenoiser.NoiseEstimation (modwt/streaming/MODWTStreamingDenoiser.java:58-65). */
typedef enum vw_noise_estimation {
    VW_NOISE_MAD = 0,  /* median(|v - median(v)|) / 0.6745 over the noise window (:195, :211-241) */
    VW_NOISE_STD = 1,  /* sample standard deviation of the noise window         (:196, :249-268) */
    VW_NOISE_FIXED = 2 /* no window, no estimate                                (:102, :197-199) */
} vw_noise_estimation;

/* State of `streams` parallel MODWTStreamingDenoiser instances that are fed blocks of equal length in
 * lock step (:50-54): the |detail| ring noiseWindow[window_size] and estimatedNoiseLevel per stream in
 * device memory, the common ring position noiseWindowIndex and samplesProcessed on the host.  Created
 * zeroed, as the Java constructor leaves them (:67-83). */
typedef struct vw_stream_state vw_stream_state;
int32_t vw_stream_state_create(vw_ctx *ctx, int64_t streams, int64_t window_size, vw_stream_state **out);
int32_t vw_stream_state_destroy(vw_ctx *ctx, vw_stream_state *state);
/* getEstimatedNoiseLevel() of every stream (:332-334): levels[streams], host memory. */
int32_t vw_stream_state_noise_levels(vw_ctx *ctx, vw_stream_state *state, double *levels);
/* Copy of the rings, window[streams][window_size] (host memory), and the ring position (may be NULL). */
int32_t vw_stream_state_window(vw_ctx *ctx, vw_stream_state *state, double *window, int64_t *window_index);
/* getSamplesProcessed() of each stream (:341-343). */
int64_t vw_stream_state_samples_processed(const vw_stream_state *state);

/* MODWTStreamingDenoiser.denoise(samples) (:93-127) for one block of b == streams rows, x [b][n]:
 * single-level forward; unless noise_estimation == VW_NOISE_FIXED the ring takes the block's |details|
 * (all of them when n <= window_size, else window_size picks stratum by stratum, :132-191) and the
 * level is re-estimated (:193-200); then
 *   |threshold_multiplier - 1| < 1e-10: WaveletDenoiser.denoise(samples, method, type) (:110);
 *   otherwise threshold = calculateThreshold(samples) * threshold_multiplier (:277-324: sigma is the
 *   estimated level, or MAD(|details|)/0.6745 of this block when there is none; UNIVERSAL sigma*sqrt(2 ln n),
 *   SURE 0.8 of that, MINIMAX piecewise, FIXED sigma) and WaveletDenoiser.denoiseFixed (:113-115).
 * VW_E_UNSUPPORTED where the reference throws: FIXED without a multiplier, BAYES with one; the state
 * has then been updated, as it has in the reference.  log_n = Math.log(n), sqrt_2_ln_n =
 * Math.sqrt(2.0 * Math.log(n)) from the host's libm (negative: the library computes them).
 * y [b][n] out (may alias x); thresholds_out (may be NULL) receives [b].  Calls on one state must not
 * overlap.  The _dev variant's workspace is vw_wavelet_denoise_workspace_bytes(b, n, 1) bytes or NULL/0. */
int32_t vw_streaming_denoise_batch(vw_ctx *ctx, vw_stream_state *state, const double *x, int64_t b, int64_t n,
                                   const double *h, int32_t lh, const double *g, int32_t lg,
                                   const double *hr, int32_t lhr, const double *gr, int32_t lgr,
                                   int32_t boundary, uint32_t flags, int32_t method, int32_t soft,
                                   double threshold_multiplier, int32_t noise_estimation, double log_n,
                                   double sqrt_2_ln_n, double *y, double *thresholds_out);
int32_t vw_streaming_denoise_batch_dev(vw_ctx *ctx, vw_stream_state *state, const double *x_dev, int64_t b,
                                       int64_t n, const double *h, int32_t lh, const double *g, int32_t lg,
                                       const double *hr, int32_t lhr, const double *gr, int32_t lgr,
                                       int32_t boundary, uint32_t flags, int32_t method, int32_t soft,
                                       double threshold_multiplier, int32_t noise_estimation, double log_n,
                                       double sqrt_2_ln_n, double *y_dev, double *thresholds_out_dev,
                                       void *workspace_dev, size_t workspace_bytes, void *stream);

/* ---- structure-of-arrays batch ---------------------------------------------- */

/* BatchSIMDMODWT.batchMODWTSoA(soaSignals, soaApprox, soaDetail, wavelet, batchSize, signalLength)
 * (modwt/BatchSIMDMODWT.java:64-81): single-level PERIODIC MODWT of `batch` signals stored sample-major,
 * soa[t * batch + b].  The reference's three code paths round differently and are all kept:
 * is_haar != 0 -> the Haar path with the literal constants 0.5 (:86-139); lh == 4 -> the four-tap path
 * (:144-211); otherwise the general path (:216-274).  Filters unscaled, as everywhere in this ABI. */
int32_t vw_batch_modwt_soa(vw_ctx *ctx, const double *soa_signals, int64_t batch, int64_t n,
                           const double *h, int32_t lh, const double *g, int32_t lg, int32_t is_haar,
                           double *soa_approx, double *soa_detail);
int32_t vw_batch_modwt_soa_dev(vw_ctx *ctx, const double *soa_signals_dev, int64_t batch, int64_t n,
                               const double *h, int32_t lh, const double *g, int32_t lg,
                               int32_t is_haar, double *soa_approx_dev, double *soa_detail_dev,
                               void *stream);

/* BatchSIMDMODWT.convertToSoA / convertFromSoA (modwt/BatchSIMDMODWT.java:282-308): out[c][r] =
 * in[r][c] for a rows x cols matrix (to SoA: rows = batch, cols = N; back: rows = N, cols = batch). */
int32_t vw_transpose(vw_ctx *ctx, const double *in, int64_t rows, int64_t cols, double *out);
int32_t vw_transpose_dev(vw_ctx *ctx, const double *in_dev, int64_t rows, int64_t cols, double *out_dev,
                         void *stream);

/* Per-signal universal threshold of an existing decomposition
 * (VectorWaveSwtAdapter.applyUniversalThreshold's threshold value,
 * swt/VectorWaveSwtAdapter.java:255-264): thresholds[b] from band d_1 of
 * coeffs [B][J+1][N]. */
int32_t vw_swt_universal_threshold_batch(vw_ctx *ctx, const double *coeffs, int64_t b, int64_t n,
                                         int32_t levels, double sqrt_2_ln_n, double *thresholds);
int32_t vw_swt_universal_threshold_batch_dev(vw_ctx *ctx, const double *coeffs_dev, int64_t b,
                                             int64_t n, int32_t levels, double sqrt_2_ln_n,
                                             double *thresholds_dev, void *stream);

/* ---- sparse detail bands ------------------------------------------------------ */

/* SWTResult.toSparse(threshold) (swt/SWTResult.java:190-222, SparseVector :262-290) for a batch of
 * decompositions coeffs [B][levels+1][N]: per detail band (row r, level j) the entries with
 * |v| > threshold, in index order.  Two calls: with indices == NULL, counts[r*levels + j] receives the
 * number of entries of each band; with indices/values given, the entries of band (r, j) are written at
 * offsets[r*levels + j] (the caller's exclusive prefix sum of the counts, B*levels + 1 entries, the last
 * one the total).  The approximation band stays
 * dense, as in the reference.  The fill pass never writes at or beyond offsets[band + 1]: if a band holds
 * more entries than its slot (the coefficients or the threshold differ from the counting pass) the surplus
 * is dropped and the host variant returns VW_E_LENGTH_MISMATCH; in the fill pass `counts`, if not NULL,
 * receives the number of entries each band really holds, which is how a _dev caller detects the same
 * condition.  The _dev variant takes device pointers for everything. */
int32_t vw_swt_sparsify_batch(vw_ctx *ctx, const double *coeffs, int64_t b, int64_t n, int32_t levels,
                              double threshold, int64_t *counts, const int64_t *offsets, int32_t *indices,
                              double *values);
int32_t vw_swt_sparsify_batch_dev(vw_ctx *ctx, const double *coeffs_dev, int64_t b, int64_t n,
                                  int32_t levels, double threshold, int64_t *counts_dev,
                                  const int64_t *offsets_dev, int32_t *indices_dev, double *values_dev,
                                  void *stream);
/* SparseSWTResult.toFull() (:225-231, SparseVector.toDense :292-298): detail bands of coeffs_out
 * [B][levels+1][N] are zeroed and the entries scattered back; offsets has B*levels + 1 entries; the
 * approximation band of coeffs_out is left untouched. */
int32_t vw_swt_densify_batch(vw_ctx *ctx, const int64_t *offsets, const int32_t *indices, const double *values,
                             int64_t b, int64_t n, int32_t levels, double *coeffs_out);
int32_t vw_swt_densify_batch_dev(vw_ctx *ctx, const int64_t *offsets_dev, const int32_t *indices_dev,
                                 const double *values_dev, int64_t b, int64_t n, int32_t levels,
                                 double *coeffs_out_dev, void *stream);

/* ---- thresholding ------------------------------------------------------- */

/* Element-wise thresholding, out of place (out may alias in).  Replaces
 * WaveletOperations.softThreshold / hardThreshold (WaveletOperations.java:
 * 89-122 -> internal/VectorOps.java:448-539) and
 * MutableMultiLevelMODWTResult.applyThreshold
 * (modwt/MutableMultiLevelMODWTResult.java:83-114).  threshold < 0 ->
 * VW_E_NOT_POSITIVE (the reference throws IllegalArgumentException).
 * NaN coefficients: the soft rule follows the reference's SCALAR form (`|c| <= thr ? 0 : signum(c)(|c| - thr)`,
 * internal/VectorOps.java:453-460 and the tail loop of the vectorised body), which lets a NaN through.  The
 * reference is not self-consistent here: its Vector-API body builds the result from `c > thr` / `c < -thr`
 * masks (:475-477), both false for a NaN, and writes 0.0 for those lanes.  For lengths the reference
 * vectorises, NaN handling is therefore unpinned; finite inputs are unaffected.  The same holds for
 * vw_wavelet_denoise_batch (WaveletDenoiser with useVectorOps). */
int32_t vw_threshold(vw_ctx *ctx, const double *in, int64_t count, double threshold, int32_t soft,
                     double *out);
int32_t vw_threshold_dev(vw_ctx *ctx, const double *in_dev, int64_t count, double threshold,
                         int32_t soft, double *out_dev, void *stream);

/* ---- raw convolutions (the reference's own seam) ------------------------ */

/* One filter over a batch of signals: out[b][t] = sum_l x[b][idx(t - l*2^(level-1))] * filter[l]
 * with the boundary rule of `boundary`; the filter is used AS GIVEN (no 1/sqrt2).
 * level = 1 is WaveletOperations.circularConvolveMODWT /
 * zeroPaddingConvolveMODWT / symmetricConvolveMODWT (WaveletOperations.java:
 * 25-49 -> internal/ScalarOps.java:671-724, 799-845); level > 1 with PERIODIC
 * is ScalarOps.circularConvolveMODWTLevel (internal/ScalarOps.java:743-789),
 * including its level - 1 >= 31 -> VW_E_TOO_LARGE check. */
int32_t vw_modwt_convolve_batch(vw_ctx *ctx, const double *x, int64_t b, int64_t n,
                                const double *filter, int32_t filter_len, int32_t level,
                                int32_t boundary, double *out);
int32_t vw_modwt_convolve_batch_dev(vw_ctx *ctx, const double *x_dev, int64_t b, int64_t n,
                                    const double *filter, int32_t filter_len, int32_t level,
                                    int32_t boundary, double *out_dev, void *stream);

/* ---- validation --------------------------------------------------------- */

/* ValidationUtils.validateFiniteValues (util/ValidationUtils.java:106-117):
 * *first_bad = index of the first NaN/Inf in x[0..count), or -1.  Returns
 * VW_E_NONFINITE when one is found. */
int32_t vw_first_non_finite(vw_ctx *ctx, const double *x, int64_t count, int64_t *first_bad);
int32_t vw_first_non_finite_dev(vw_ctx *ctx, const double *x_dev, int64_t count,
                                int64_t *first_bad_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VWGPU_H */
