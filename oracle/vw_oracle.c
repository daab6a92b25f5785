/*
 * vw_oracle.c -- CPU restatement of the reference's MODWT / SWT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library calls, links or
 * loads this file; it exists so the HIP kernels can be checked against an
 * independent statement of the reference's arithmetic.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * The reference (Prophetizo/VectorWave) is Java; a JVM is not available in the
 * build image, so every function below restates one Java method and cites it
 * (paths relative to src/main/java/ai/prophetizo/wavelet/).  Java evaluates
 * `sum += a * b` as a rounded multiply followed by a rounded add, never as a
 * fused multiply-add, so this file MUST be compiled with -ffp-contract=off and
 * without -ffast-math.  Loops keep the reference's summation order.
 *
 * Parity pinning: the closed-form known-answer vectors in the reference's own
 * unit tests are replayed against this file by tests/test_oracle_kat.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VWO_PERIODIC 0
#define VWO_ZERO_PADDING 1
#define VWO_SYMMETRIC 2

#define VWO_MAX_DECOMPOSITION_LEVELS 10 /* modwt/MultiLevelMODWTTransform.java:101 */
#define VWO_MAX_SAFE_SHIFT_BITS 31      /* modwt/MultiLevelMODWTTransform.java:114 */

/* ------------------------------------------------------------------------ */
/* index helpers                                                            */
/* ------------------------------------------------------------------------ */

/* util/MathUtils.java:31-52 */
static int64_t symmetric_boundary_extension(int64_t idx, int64_t n) {
    if (idx >= 0 && idx < n) return idx;
    int64_t period = 2 * n;
    idx = ((idx % period) + period) % period;
    if (idx >= n) idx = period - idx - 1;
    return idx;
}

/* internal/ScalarOps.java:704-717 */
static int64_t circular_index(int64_t idx, int64_t n) {
    if (idx >= 0 && idx < n) return idx;
    if (idx < 0 && idx >= -n) return idx + n;
    return ((idx % n) + n) % n;
}

/* ------------------------------------------------------------------------ */
/* a1..a4: the three convolutions + the dilated form                        */
/* ------------------------------------------------------------------------ */

/* internal/ScalarOps.java:694-724 (circularConvolveMODWTScalar) */
void vwo_circular_convolve_scalar(const double *signal, int64_t n, const double *filter,
                                  int64_t flen, double *output) {
    for (int64_t t = 0; t < n; t++) {
        double sum = 0.0;
        for (int64_t l = 0; l < flen; l++) {
            sum += signal[circular_index(t - l, n)] * filter[l];
        }
        output[t] = sum;
    }
}

/* internal/VectorOps.java:554-649 (circularConvolveMODWTVectorized): tap-outer
 * accumulation into a cleared output, taps equal to 0.0 skipped.  Per output
 * element the products are added in the same ascending-l order as the scalar
 * loop, so the values are identical; kept as a separate function because it is
 * what the reference executes for N >= 128 and what the CPU baseline times. */
void vwo_circular_convolve_vector_order(const double *signal, int64_t n, const double *filter,
                                        int64_t flen, double *output) {
    for (int64_t t = 0; t < n; t++) output[t] = 0.0;
    for (int64_t l = 0; l < flen; l++) {
        double f = filter[l];
        if (f == 0.0) continue;
        int64_t head = l < n ? l : n; /* outputs whose source index wraps */
        for (int64_t t = 0; t < head; t++) output[t] += signal[circular_index(t - l, n)] * f;
        for (int64_t t = head; t < n; t++) output[t] += signal[t - l] * f;
    }
}

/* internal/ScalarOps.java:671-677 + :878-891 (dispatch with the Vector API on) */
void vwo_circular_convolve(const double *signal, int64_t n, const double *filter, int64_t flen,
                           double *output) {
    int vector = 0;
    if (n >= 32) vector = (n >= 128) || (flen <= 8);
    if (vector)
        vwo_circular_convolve_vector_order(signal, n, filter, flen, output);
    else
        vwo_circular_convolve_scalar(signal, n, filter, flen, output);
}

/* internal/ScalarOps.java:743-789 (circularConvolveMODWTLevel); returns 5
 * (VAL_TOO_LARGE) when level-1 >= 31, else 0. */
int vwo_circular_convolve_level(const double *signal, int64_t n, const double *filter,
                                int64_t flen, double *output, int level) {
    if (level - 1 >= VWO_MAX_SAFE_SHIFT_BITS) return 5;
    int64_t shift = (int64_t)1 << (level - 1);
    for (int64_t t = 0; t < n; t++) {
        double sum = 0.0;
        for (int64_t l = 0; l < flen; l++) {
            sum += signal[circular_index(t - shift * l, n)] * filter[l];
        }
        output[t] = sum;
    }
    return 0;
}

/* internal/ScalarOps.java:799-817 */
void vwo_zero_padding_convolve(const double *signal, int64_t n, const double *filter,
                               int64_t flen, double *output) {
    for (int64_t t = 0; t < n; t++) {
        double sum = 0.0;
        for (int64_t l = 0; l < flen; l++) {
            int64_t idx = t - l;
            if (idx >= 0 && idx < n) sum += signal[idx] * filter[l];
        }
        output[t] = sum;
    }
}

/* internal/ScalarOps.java:827-845 */
void vwo_symmetric_convolve(const double *signal, int64_t n, const double *filter, int64_t flen,
                            double *output) {
    for (int64_t t = 0; t < n; t++) {
        double sum = 0.0;
        for (int64_t l = 0; l < flen; l++) {
            sum += signal[symmetric_boundary_extension(t - l, n)] * filter[l];
        }
        output[t] = sum;
    }
}

/* modwt/MultiLevelMODWTTransform.java:840-864 (circularConvolveMODWTDirect) */
static void circular_convolve_direct(const double *signal, int64_t n, const double *filter,
                                     int64_t flen, double *output) {
    int64_t eff = flen < n ? flen : n;
    for (int64_t t = 0; t < n; t++) {
        double sum = 0.0;
        int64_t maxk = eff < t + 1 ? eff : t + 1;
        for (int64_t k = 0; k < maxk; k++) sum += filter[k] * signal[t - k];
        for (int64_t k = maxk; k < eff; k++) sum += filter[k] * signal[t - k + n];
        output[t] = sum;
    }
}

static void convolve_by_mode(const double *signal, int64_t n, const double *filter, int64_t flen,
                             int boundary, double *output) {
    if (boundary == VWO_PERIODIC)
        vwo_circular_convolve(signal, n, filter, flen, output);
    else if (boundary == VWO_ZERO_PADDING)
        vwo_zero_padding_convolve(signal, n, filter, flen, output);
    else
        vwo_symmetric_convolve(signal, n, filter, flen, output);
}

/* ------------------------------------------------------------------------ */
/* a5 / a6: single-level transform                                          */
/* ------------------------------------------------------------------------ */

static double *scaled_copy(const double *taps, int len) {
    /* modwt/MODWTTransform.java:133-142: scale = 1.0 / Math.sqrt(2.0); tap * scale */
    double scale = 1.0 / sqrt(2.0);
    double *out = (double *)malloc(sizeof(double) * (size_t)(len > 0 ? len : 1));
    for (int i = 0; i < len; i++) out[i] = taps[i] * scale;
    return out;
}

/* modwt/MODWTTransform.java:123-179 (forward); h/g are the wavelet's UNSCALED
 * decomposition filters. */
void vwo_modwt_forward(const double *x, int64_t n, const double *h, int lh, const double *g,
                       int lg, int boundary, double *approx, double *detail) {
    double *sh = scaled_copy(h, lh), *sg = scaled_copy(g, lg);
    convolve_by_mode(x, n, sh, lh, boundary, approx);
    convolve_by_mode(x, n, sg, lg, boundary, detail);
    free(sh);
    free(sg);
}

/* modwt/MODWTTransform.java:193-291 (inverse) and :592-662 (inverseBatchOptimized).
 * The loop bound is the low-pass reconstruction length and the high-pass filter
 * is indexed with the same l (the reference does this even when the lengths
 * differ; callers must pass lgr >= lhr).  `batch_variant` selects the SYMMETRIC
 * flavour: 0 = inverse() reflects t-l (:269-287), 1 = inverseBatchOptimized()
 * reflects t+l (:644-657). */
void vwo_modwt_inverse(const double *approx, const double *detail, int64_t n, const double *hr,
                       int lhr, const double *gr, int lgr, int boundary, int batch_variant,
                       double *out) {
    double *lo = scaled_copy(hr, lhr), *hi = scaled_copy(gr, lgr);
    for (int64_t t = 0; t < n; t++) {
        double sum = 0.0;
        for (int l = 0; l < lhr; l++) {
            int64_t idx;
            if (boundary == VWO_PERIODIC) {
                idx = (t + l) % n;
            } else if (boundary == VWO_ZERO_PADDING) {
                idx = t + l;
                if (idx >= n) continue;
            } else if (!batch_variant) {
                idx = symmetric_boundary_extension(t - l, n);
            } else {
                int64_t period = n << 1;
                int64_t mod = (t + l) % period;
                idx = mod < n ? mod : period - mod - 1;
            }
            sum += lo[l] * approx[idx] + hi[l] * detail[idx];
        }
        out[t] = sum;
    }
    free(lo);
    free(hi);
}

/* ------------------------------------------------------------------------ */
/* a7 / a8: multi-level cascade                                             */
/* ------------------------------------------------------------------------ */

/* modwt/MultiLevelMODWTTransform.java:477-523 (calculateMaxLevels) */
int vwo_max_levels(int64_t n, int filter_len) {
    if (n <= filter_len) return 0;
    int max_level = 1;
    int64_t lm1 = filter_len - 1;
    while (max_level < VWO_MAX_DECOMPOSITION_LEVELS) {
        if (max_level - 1 >= VWO_MAX_SAFE_SHIFT_BITS) break;
        int64_t scaled = lm1 * ((int64_t)1 << (max_level - 1)) + 1;
        if (scaled > n) break;
        max_level++;
    }
    return max_level - 1;
}

/* modwt/MultiLevelMODWTTransform.java:697-752 and :560-615: zero-stuff by
 * 2^(level-1) and scale every tap by 1/sqrt(2).  Returns malloc'd filter. */
static double *stuffed_filter(const double *taps, int len, int level, int64_t *out_len) {
    double scale = 1.0 / sqrt(2.0);
    int64_t up = (int64_t)1 << (level - 1);
    int64_t flen = (int64_t)(len - 1) * up + 1;
    double *f = (double *)calloc((size_t)flen, sizeof(double));
    for (int i = 0; i < len; i++) f[(int64_t)i * up] = taps[i] * scale;
    *out_len = flen;
    return f;
}

/* modwt/MultiLevelMODWTTransform.java:793-834 (applyScaledMODWT) */
static void apply_scaled_modwt(const double *signal, int64_t n, const double *lo, int64_t lo_len,
                               const double *hi, int64_t hi_len, int boundary, double *approx,
                               double *detail) {
    /* :801-806 truncation (getTruncatedFilter keeps the first n entries) */
    if (lo_len > n) lo_len = n;
    if (hi_len > n) hi_len = n;
    if (boundary == VWO_PERIODIC) {
        if (n < 64 || lo_len > n / 2) {
            circular_convolve_direct(signal, n, lo, lo_len, approx);
            circular_convolve_direct(signal, n, hi, hi_len, detail);
        } else {
            vwo_circular_convolve(signal, n, lo, lo_len, approx);
            vwo_circular_convolve(signal, n, hi, hi_len, detail);
        }
    } else if (boundary == VWO_ZERO_PADDING) {
        vwo_zero_padding_convolve(signal, n, lo, lo_len, approx);
        vwo_zero_padding_convolve(signal, n, hi, hi_len, detail);
    } else {
        vwo_symmetric_convolve(signal, n, lo, lo_len, approx);
        vwo_symmetric_convolve(signal, n, hi, hi_len, detail);
    }
}

/* modwt/MultiLevelMODWTTransform.java:214-268 (decompose) / :299-353.
 * coeffs is [levels+1][n]: d_1 .. d_J, a_J.  Returns 0, or 104
 * (CFG_INVALID_DECOMPOSITION_LEVEL) / 6 (VAL_EMPTY) like the reference's checks. */
int vwo_multilevel_decompose(const double *x, int64_t n, int levels, const double *h, int lh,
                             const double *g, int lg, int boundary, double *coeffs) {
    if (n == 0) return 6;
    int max_levels = vwo_max_levels(n, lh);
    if (levels < 1 || levels > max_levels) return 104;
    double *cur = (double *)malloc(sizeof(double) * (size_t)n);
    double *next = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(cur, x, sizeof(double) * (size_t)n);
    for (int level = 1; level <= levels; level++) {
        int64_t lo_len, hi_len;
        double *lo = stuffed_filter(h, lh, level, &lo_len);
        double *hi = stuffed_filter(g, lg, level, &hi_len);
        apply_scaled_modwt(cur, n, lo, lo_len, hi, hi_len, boundary, next,
                           coeffs + (int64_t)(level - 1) * n);
        free(lo);
        free(hi);
        double *tmp = cur;
        cur = next;
        next = tmp;
    }
    memcpy(coeffs + (int64_t)levels * n, cur, sizeof(double) * (size_t)n);
    free(cur);
    free(next);
    return 0;
}

/* modwt/MultiLevelMODWTTransform.java:621-689 (applyScaledInverseMODWT), dense
 * over the zero-stuffed filters exactly as the reference iterates them. */
static void apply_scaled_inverse(const double *approx, const double *details, int64_t n,
                                 const double *lo, int64_t lo_len, const double *hi,
                                 int64_t hi_len, int boundary, double *out) {
    if (lo_len > n) lo_len = n; /* :628-636 truncation */
    if (hi_len > n) hi_len = n;
    if (boundary == VWO_PERIODIC) {
        for (int64_t t = 0; t < n; t++) {
            double sum = 0.0;
            for (int64_t l = 0; l < lo_len; l++) {
                int64_t idx = (t + l) % n;
                sum += lo[l] * approx[idx] + hi[l] * details[idx];
            }
            out[t] = sum;
        }
    } else if (boundary == VWO_ZERO_PADDING) {
        for (int64_t t = 0; t < n; t++) {
            double sum = 0.0;
            for (int64_t l = 0; l < lo_len; l++) {
                int64_t idx = t + l;
                if (idx < n) sum += lo[l] * approx[idx] + hi[l] * details[idx];
            }
            out[t] = sum;
        }
    } else {
        int64_t max_len = lo_len > hi_len ? lo_len : hi_len;
        for (int64_t t = 0; t < n; t++) {
            double sum = 0.0;
            for (int64_t l = 0; l < max_len; l++) {
                int64_t idx = symmetric_boundary_extension(t - l, n);
                if (l < lo_len) sum += lo[l] * approx[idx];
                if (l < hi_len) sum += hi[l] * details[idx];
            }
            out[t] = sum;
        }
    }
}

/* modwt/MultiLevelMODWTTransform.java:362-375 (reconstruct).  hr/gr: UNSCALED
 * reconstruction filters; for PERIODIC / ZERO_PADDING the reference indexes
 * gr[l] for l < len(hr-stuffed), so lgr >= lhr is required (equal for every
 * orthogonal wavelet). */
void vwo_multilevel_reconstruct(const double *coeffs, int64_t n, int levels, const double *hr,
                                int lhr, const double *gr, int lgr, int boundary, double *out) {
    double *rec = (double *)malloc(sizeof(double) * (size_t)n);
    double *next = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(rec, coeffs + (int64_t)levels * n, sizeof(double) * (size_t)n);
    for (int level = levels; level >= 1; level--) {
        int64_t lo_len, hi_len;
        double *lo = stuffed_filter(hr, lhr, level, &lo_len);
        double *hi = stuffed_filter(gr, lgr, level, &hi_len);
        apply_scaled_inverse(rec, coeffs + (int64_t)(level - 1) * n, n, lo, lo_len, hi, hi_len,
                             boundary, next);
        free(lo);
        free(hi);
        double *tmp = rec;
        rec = next;
        next = tmp;
    }
    memcpy(out, rec, sizeof(double) * (size_t)n);
    free(rec);
    free(next);
}

/* ------------------------------------------------------------------------ */
/* a10 / a11: thresholds, universal threshold, SWT denoise                  */
/* ------------------------------------------------------------------------ */

/* modwt/MutableMultiLevelMODWTResult.java:97-114 (applyThresholdToArray), in place */
void vwo_apply_threshold(double *c, int64_t n, double threshold, int soft) {
    for (int64_t i = 0; i < n; i++) {
        double a = fabs(c[i]);
        if (soft) {
            if (a > threshold) {
                double sg = c[i] > 0.0 ? 1.0 : (c[i] < 0.0 ? -1.0 : c[i]); /* Math.signum */
                c[i] = sg * (a - threshold);
            } else {
                c[i] = 0.0;
            }
        } else if (a <= threshold) {
            c[i] = 0.0;
        }
    }
}

/* internal/VectorOps.java:448-497 (scalar branch :454-459), out of place */
void vwo_soft_threshold(const double *c, int64_t n, double threshold, double *out) {
    for (int64_t i = 0; i < n; i++) {
        double a = fabs(c[i]);
        double sg = c[i] > 0.0 ? 1.0 : (c[i] < 0.0 ? -1.0 : c[i]);
        out[i] = a <= threshold ? 0.0 : sg * (a - threshold);
    }
}

/* internal/VectorOps.java:502-539 */
void vwo_hard_threshold(const double *c, int64_t n, double threshold, double *out) {
    for (int64_t i = 0; i < n; i++) out[i] = fabs(c[i]) <= threshold ? 0.0 : c[i];
}

static int cmp_double(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* swt/VectorWaveSwtAdapter.java:377-395 (estimateNoiseSigma) */
double vwo_estimate_noise_sigma(const double *coeffs, int64_t n) {
    double *a = (double *)malloc(sizeof(double) * (size_t)n);
    for (int64_t i = 0; i < n; i++) a[i] = fabs(coeffs[i]);
    qsort(a, (size_t)n, sizeof(double), cmp_double);
    double median;
    if (n % 2 == 0)
        median = (a[n / 2 - 1] + a[n / 2]) / 2.0;
    else
        median = a[n / 2];
    free(a);
    return median / 0.6745;
}

/* swt/VectorWaveSwtAdapter.java:255-270: threshold = sigma * Math.sqrt(2 * Math.log(n)) */
double vwo_universal_threshold(const double *finest_detail, int64_t n) {
    double sigma = vwo_estimate_noise_sigma(finest_detail, n);
    return sigma * sqrt(2 * log((double)n));
}

/* swt/VectorWaveSwtAdapter.java:296-312 (denoise).  threshold < 0 selects the
 * universal threshold.  Returns the status of the decomposition. */
int vwo_swt_denoise(const double *x, int64_t n, int levels, const double *h, int lh,
                    const double *g, int lg, const double *hr, int lhr, const double *gr, int lgr,
                    int boundary, double threshold, int soft, double *out) {
    double *coeffs = (double *)malloc(sizeof(double) * (size_t)n * (size_t)(levels + 1));
    int st = vwo_multilevel_decompose(x, n, levels, h, lh, g, lg, boundary, coeffs);
    if (st != 0) {
        free(coeffs);
        return st;
    }
    double thr = threshold < 0 ? vwo_universal_threshold(coeffs, n) : threshold;
    for (int level = 1; level <= levels; level++)
        vwo_apply_threshold(coeffs + (int64_t)(level - 1) * n, n, thr, soft);
    vwo_multilevel_reconstruct(coeffs, n, levels, hr, lhr, gr, lgr, boundary, out);
    free(coeffs);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* denoising/WaveletDenoiser.java: threshold selection + single-/multi-level denoise */
/* ------------------------------------------------------------------------ */

/* denoising/WaveletDenoiser.java:468-483 (calculateSURERisk) */
static double sure_risk(const double *coeffs, int64_t n, double threshold, double sigma) {
    double sigma2 = sigma * sigma;
    double risk = -(double)n * sigma2;
    for (int64_t i = 0; i < n; i++) {
        double c = coeffs[i];
        double absC = fabs(c);
        if (absC <= threshold) {
            risk += c * c;
        } else {
            risk += sigma2 + (absC - threshold) * (absC - threshold);
        }
    }
    return risk / (double)n;
}

/* denoising/WaveletDenoiser.java:385-427 (calculateThreshold) with :432-463 (SURE),
 * :488-499 (MINIMAX), :512-539 (BAYES).  method: 0 UNIVERSAL, 1 SURE, 2 MINIMAX, 3 BAYES.
 * Returns NaN for FIXED (4) and unknown methods, where the reference throws. */
double vwo_denoiser_threshold(const double *coeffs, int64_t n, double sigma, int method) {
    switch (method) {
        case 0:
            return sigma * sqrt(2.0 * log((double)n));
        case 1: {
            double *sorted = (double *)malloc(sizeof(double) * (size_t)n);
            for (int64_t i = 0; i < n; i++) sorted[i] = fabs(coeffs[i]);
            qsort(sorted, (size_t)n, sizeof(double), cmp_double);
            double min_risk = INFINITY, best = 0;
            for (int64_t k = 0; k < n; k++) {
                double t = sorted[k];
                double risk = sure_risk(coeffs, n, t, sigma);
                if (risk < min_risk) {
                    min_risk = risk;
                    best = t;
                }
            }
            free(sorted);
            double universal = sigma * sqrt(2.0 * log((double)n));
            if (best > universal) best = universal;
            return best;
        }
        case 2: {
            double logN = log((double)n);
            if (n <= 32) return 0;
            if (n <= 64) return sigma * 0.3936 + 0.1829 * sigma * logN;
            return sigma * (0.4745 + 0.1148 * logN);
        }
        case 3: {
            double sigma2 = sigma * sigma;
            double mean = 0.0;
            for (int64_t i = 0; i < n; i++) mean += coeffs[i];
            mean /= (double)n;
            double variance = 0.0;
            for (int64_t i = 0; i < n; i++) {
                double diff = coeffs[i] - mean;
                variance += diff * diff;
            }
            variance /= (double)n;
            double d = variance - sigma2;
            double sigmaX2 = d > 0.0 ? d : 0.0; /* Math.max(0.0, variance - sigma2) for finite input */
            double sigmaX = sqrt(sigmaX2 + 1e-10);
            return sigma2 / sigmaX;
        }
        default:
            return NAN;
    }
}

/* denoising/WaveletDenoiser.java:544-568 (applyThreshold), out of place */
static void denoiser_apply_threshold(const double *c, int64_t n, double threshold, int soft, double *out) {
    if (soft)
        vwo_soft_threshold(c, n, threshold, out);
    else
        vwo_hard_threshold(c, n, threshold, out);
}

/* denoising/WaveletDenoiser.java:121-139 (denoise) and :350-360 (denoiseFixed, method == 4):
 * single-level MODWTTransform.forward -> threshold the detail band -> MODWTTransform.inverse */
int vwo_wavelet_denoise(const double *x, int64_t n, const double *h, int lh, const double *g, int lg,
                        const double *hr, int lhr, const double *gr, int lgr, int boundary, int method,
                        double fixed_threshold, int soft, double *out, double *threshold_out) {
    double *approx = (double *)malloc(sizeof(double) * (size_t)n * 3);
    double *detail = approx + n, *den = detail + n;
    vwo_modwt_forward(x, n, h, lh, g, lg, boundary, approx, detail);
    double threshold;
    if (method == 4) {
        threshold = fixed_threshold;
    } else {
        double sigma = vwo_estimate_noise_sigma(detail, n);
        threshold = vwo_denoiser_threshold(detail, n, sigma, method);
    }
    if (threshold_out) *threshold_out = threshold;
    denoiser_apply_threshold(detail, n, threshold, soft, den);
    vwo_modwt_inverse(approx, den, n, hr, lhr, gr, lgr, boundary, 0, out);
    free(approx);
    return 0;
}

/* denoising/WaveletDenoiser.java:152-167 + DenoisedMultiLevelResult :179-227: sigma from level 1,
 * per level threshold = calculateThreshold(details_j, sigma / Math.sqrt(1 << (j-1)), method) */
int vwo_wavelet_denoise_multilevel(const double *x, int64_t n, int levels, const double *h, int lh,
                                   const double *g, int lg, const double *hr, int lhr, const double *gr,
                                   int lgr, int boundary, int method, int soft, double *out,
                                   double *thresholds_out /* [levels] or NULL */) {
    double *coeffs = (double *)malloc(sizeof(double) * (size_t)n * (size_t)(levels + 1));
    int st = vwo_multilevel_decompose(x, n, levels, h, lh, g, lg, boundary, coeffs);
    if (st != 0) {
        free(coeffs);
        return st;
    }
    double sigma = vwo_estimate_noise_sigma(coeffs, n);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)n);
    for (int level = 1; level <= levels; level++) {
        double *d = coeffs + (int64_t)(level - 1) * n;
        double level_scale = sqrt((double)(1 << (level - 1)));
        double threshold = vwo_denoiser_threshold(d, n, sigma / level_scale, method);
        if (thresholds_out) thresholds_out[level - 1] = threshold;
        denoiser_apply_threshold(d, n, threshold, soft, tmp);
        memcpy(d, tmp, sizeof(double) * (size_t)n);
    }
    free(tmp);
    vwo_multilevel_reconstruct(coeffs, n, levels, hr, lhr, gr, lgr, boundary, out);
    free(coeffs);
    return 0;
}

/* rows in parallel; levels == 0 selects the single-level form */
int vwo_wavelet_denoise_batch(const double *x, int64_t b, int64_t n, int levels, const double *h, int lh,
                              const double *g, int lg, const double *hr, int lhr, const double *gr,
                              int lgr, int boundary, int method, double fixed_threshold, int soft,
                              double *out, double *thresholds_out /* [b][max(levels,1)] or NULL */) {
    int status = 0;
    const int nl = levels > 0 ? levels : 1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < b; i++) {
        int st;
        if (levels == 0)
            st = vwo_wavelet_denoise(x + i * n, n, h, lh, g, lg, hr, lhr, gr, lgr, boundary, method,
                                     fixed_threshold, soft, out + i * n,
                                     thresholds_out ? thresholds_out + i * nl : NULL);
        else
            st = vwo_wavelet_denoise_multilevel(x + i * n, n, levels, h, lh, g, lg, hr, lhr, gr, lgr,
                                                boundary, method, soft, out + i * n,
                                                thresholds_out ? thresholds_out + i * nl : NULL);
        if (st != 0) {
#pragma omp critical
            status = st;
        }
    }
    return status;
}

/* ------------------------------------------------------------------------ */
/* modwt/streaming/MODWTStreamingDenoiser.java: block-by-block denoiser with a noise window */
/* ------------------------------------------------------------------------ */
/* The reference's tests for this class hold properties only (lengths, "noise level is positive",
 * "multipliers change the result"), no golden values: parity of this part is UNPINNED beyond the
 * single-level WaveletDenoiser path underneath it, which is pinned. */

/* util/MathUtils.java:95-112 (median): an order statistic, so a sort gives what quickSelect gives */
static double median_of(const double *v, int64_t n) {
    double *a = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(a, v, sizeof(double) * (size_t)n);
    qsort(a, (size_t)n, sizeof(double), cmp_double);
    double m = (n % 2 == 1) ? a[n / 2] : (a[n / 2 - 1] + a[n / 2]) / 2.0;
    free(a);
    return m;
}

/* MODWTStreamingDenoiser.java:211-241 (calculateMAD) over util/MathUtils.java:122-138
 * (medianAbsoluteDeviation): median(|v_i - median(v)|); 0 when no entry is finite or every finite
 * entry is zero.  Meaningful for finite input only (an order statistic over NaN has no defined value). */
double vwo_stream_mad(const double *values, int64_t n) {
    int64_t valid = 0;
    int has_data = 0;
    for (int64_t i = 0; i < n; i++)
        if (isfinite(values[i])) {
            valid++;
            if (values[i] != 0.0) has_data = 1;
        }
    if (valid == 0 || !has_data) return 0.0;
    double med = median_of(values, n);
    double *dev = (double *)malloc(sizeof(double) * (size_t)n);
    for (int64_t i = 0; i < n; i++) dev[i] = fabs(values[i] - med);
    double mad = median_of(dev, n);
    free(dev);
    return mad;
}

/* MODWTStreamingDenoiser.java:249-268 (calculateSTD) over util/MathUtils.java:234-258 */
double vwo_stream_std(const double *values, int64_t n) {
    int64_t valid = 0;
    for (int64_t i = 0; i < n; i++)
        if (isfinite(values[i])) valid++;
    if (valid < 2) return 0.0;
    double sum = 0.0;
    for (int64_t i = 0; i < n; i++) sum += values[i];
    double mean = sum / (double)n;
    double ssd = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double diff = values[i] - mean;
        ssd += diff * diff;
    }
    return sqrt(ssd / (double)(n - 1));
}

/* MODWTStreamingDenoiser.java:132-201 (updateNoiseEstimation): writes |detail| samples into the ring
 * `window` (all of them when there are at most `w`, else a stratified pick of exactly `w`), then
 * re-estimates the noise level.  estimation: 0 MAD, 1 STD. */
void vwo_stream_update_noise(const double *details, int64_t n, double *window, int64_t w, int64_t *index,
                             int estimation, double *level) {
    int64_t wi = *index;
    if (n <= w) {
        for (int64_t i = 0; i < n; i++) {
            window[wi] = fabs(details[i]);
            wi = (wi + 1) % w;
        }
    } else {
        int64_t strata = w < 10 ? w : 10;
        int64_t per = w / strata, extra = w % strata, size = n / strata;
        int64_t taken = 0;
        for (int64_t s = 0; s < strata; s++) {
            int64_t start = s * size;
            int64_t end = (s == strata - 1) ? n : (s + 1) * size;
            int64_t take = per + (s < extra ? 1 : 0);
            if (take > 0) {
                int64_t len = end - start;
                int64_t step = len / take > 1 ? len / take : 1;
                for (int64_t i = 0; i < take && taken < w; i++) {
                    int64_t idx = start + (i * step) % len;
                    if (idx < n) {
                        window[wi] = fabs(details[idx]);
                        wi = (wi + 1) % w;
                        taken++;
                    }
                }
            }
        }
        int64_t remaining = w - taken;
        if (remaining > 0) {
            int64_t from = n - remaining > 0 ? n - remaining : 0;
            for (int64_t i = from; i < n && taken < w; i++) {
                window[wi] = fabs(details[i]);
                wi = (wi + 1) % w;
                taken++;
            }
        }
    }
    *index = wi;
    if (estimation == 0)
        *level = vwo_stream_mad(window, w) / 0.6745;
    else if (estimation == 1)
        *level = vwo_stream_std(window, w);
}

/* MODWTStreamingDenoiser.java:277-324 (calculateThreshold) given the detail band of the block.
 * method: 0 UNIVERSAL, 1 SURE, 2 MINIMAX, 4 FIXED; NaN for BAYES (3) and unknown, where the reference
 * throws IllegalArgumentException. */
double vwo_stream_base_threshold(const double *details, int64_t n, double level, int estimation, int method) {
    double sigma = level;
    if (sigma <= 0.0 || estimation == 2) {
        double *a = (double *)malloc(sizeof(double) * (size_t)n);
        for (int64_t i = 0; i < n; i++) a[i] = fabs(details[i]);
        sigma = vwo_stream_mad(a, n) / 0.6745;
        free(a);
    }
    switch (method) {
        case 0:
            return sigma * sqrt(2.0 * log((double)n));
        case 1:
            return sigma * sqrt(2.0 * log((double)n)) * 0.8;
        case 2: {
            double logN = log((double)n);
            if (n <= 32) return 0.0;
            if (n <= 64) return sigma * (0.3936 + 0.1829 * logN);
            return sigma * (0.4745 + 0.1148 * logN);
        }
        case 4:
            return sigma;
        default:
            return NAN;
    }
}

/* MODWTStreamingDenoiser.java:93-127 (denoise) for one block of one stream.  State: window[w], *index,
 * *level (all updated in place; untouched when estimation == 2, FIXED).  Returns 0, or 105 where the
 * reference throws for the method (FIXED without a multiplier; BAYES or unknown with one). */
int vwo_streaming_denoise(const double *x, int64_t n, const double *h, int lh, const double *g, int lg,
                          const double *hr, int lhr, const double *gr, int lgr, int boundary, int method,
                          int soft, double multiplier, int estimation, double *window, int64_t w,
                          int64_t *index, double *level, double *out, double *threshold_out) {
    double *approx = (double *)malloc(sizeof(double) * (size_t)n * 2);
    double *detail = approx + n;
    vwo_modwt_forward(x, n, h, lh, g, lg, boundary, approx, detail);
    if (estimation != 2) vwo_stream_update_noise(detail, n, window, w, index, estimation, level);
    int st;
    if (fabs(multiplier - 1.0) < 1e-10) {
        if (method < 0 || method > 3) {
            st = 105;
        } else {
            st = vwo_wavelet_denoise(x, n, h, lh, g, lg, hr, lhr, gr, lgr, boundary, method, 0.0, soft, out,
                                     threshold_out);
        }
    } else {
        double base = vwo_stream_base_threshold(detail, n, *level, estimation, method);
        if (method == 3 || method < 0 || method > 4) {
            st = 105;
        } else {
            double adjusted = base * multiplier;
            st = vwo_wavelet_denoise(x, n, h, lh, g, lg, hr, lhr, gr, lgr, boundary, 4, adjusted, soft, out,
                                     threshold_out);
        }
    }
    free(approx);
    return st;
}

/* b independent streams, one block each: x[b][n], window[b][w], level[b]; the ring position is common
 * to the streams because every stream receives blocks of the same length */
int vwo_streaming_denoise_batch(const double *x, int64_t b, int64_t n, const double *h, int lh,
                                const double *g, int lg, const double *hr, int lhr, const double *gr,
                                int lgr, int boundary, int method, int soft, double multiplier,
                                int estimation, double *window, int64_t w, int64_t *index, double *level,
                                double *out, double *thresholds_out /* [b] or NULL */) {
    int status = 0;
    int64_t next = *index;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < b; i++) {
        int64_t wi = *index;
        int st = vwo_streaming_denoise(x + i * n, n, h, lh, g, lg, hr, lhr, gr, lgr, boundary, method, soft,
                                       multiplier, estimation, window + i * w, w, &wi, level + i,
                                       out + i * n, thresholds_out ? thresholds_out + i : NULL);
        if (i == 0) {
#pragma omp critical
            next = wi;
        }
        if (st != 0) {
#pragma omp critical
            status = st;
        }
    }
    *index = next;
    return status;
}

/* ------------------------------------------------------------------------ */
/* modwt/streaming/MODWTStreamingTransformImpl.java: sliding-window block transform */
/* ------------------------------------------------------------------------ */

/* State of one MODWTStreamingTransformImpl (:54-60): the circular buffer of bufferSize + overlap samples
 * (overlap = filter length - 1, :93-94), the write position and the count of unconsumed samples. */
typedef struct {
    double *ring;
    int64_t ring_len, buffer_size, overlap, write_pos, in_buffer;
} vwo_stream_xform;

vwo_stream_xform *vwo_stream_xform_create(int64_t buffer_size, int filter_len) {
    vwo_stream_xform *s = (vwo_stream_xform *)malloc(sizeof(*s));
    s->buffer_size = buffer_size;
    s->overlap = filter_len - 1;
    s->ring_len = buffer_size + s->overlap;
    s->ring = (double *)calloc((size_t)s->ring_len, sizeof(double));
    s->write_pos = 0;
    s->in_buffer = 0;
    return s;
}

void vwo_stream_xform_destroy(vwo_stream_xform *s) {
    if (s) {
        free(s->ring);
        free(s);
    }
}

int64_t vwo_stream_xform_level(const vwo_stream_xform *s) { return s->in_buffer; }

/* :188-203 (reset) */
void vwo_stream_xform_reset(vwo_stream_xform *s) {
    memset(s->ring, 0, sizeof(double) * (size_t)s->ring_len);
    s->write_pos = 0;
    s->in_buffer = 0;
}

/* :145-188 (process -> processSampleInternal -> processBuffer), sample by sample: every time
 * bufferSize unconsumed samples are buffered, the last bufferSize samples are transformed (single level)
 * and bufferSize - overlap of them count as consumed.  Blocks go to out[block][2][bufferSize] = (detail,
 * approx); returns the number of blocks emitted (at most max_blocks are written). */
int64_t vwo_stream_xform_process(vwo_stream_xform *s, const double *data, int64_t len, const double *h,
                                 int lh, const double *g, int lg, int boundary, double *out,
                                 int64_t max_blocks) {
    const int64_t bsz = s->buffer_size;
    int64_t blocks = 0;
    double *buf = (double *)malloc(sizeof(double) * (size_t)bsz);
    for (int64_t t = 0; t < len; t++) {
        s->ring[s->write_pos] = data[t];
        s->write_pos = (s->write_pos + 1) % s->ring_len;
        s->in_buffer++;
        if (s->in_buffer >= bsz) {
            int64_t read_pos = (s->write_pos - bsz + s->ring_len) % s->ring_len;
            for (int64_t i = 0; i < bsz; i++) buf[i] = s->ring[(read_pos + i) % s->ring_len];
            if (blocks < max_blocks)
                vwo_modwt_forward(buf, bsz, h, lh, g, lg, boundary, out + (blocks * 2 + 1) * bsz,
                                  out + (blocks * 2) * bsz);
            blocks++;
            s->in_buffer -= bsz - s->overlap;
        }
    }
    free(buf);
    return blocks;
}

/* The same walk with the reference's failure rule: MODWTTransform.forward validates its window
 * (modwt/MODWTTransform.java:353-357), so the first window that holds a NaN or an infinity throws out of
 * processBuffer (:194) before the consumed samples are subtracted (:226) and before the sample is counted
 * (:179); the rest of `data` is never fed.  *failed_at receives the index of the sample whose window failed
 * (-1 if none); the state is left as the reference leaves it (sample stored, count not reduced). */
int64_t vwo_stream_xform_process_checked(vwo_stream_xform *s, const double *data, int64_t len, const double *h,
                                         int lh, const double *g, int lg, int boundary, double *out,
                                         int64_t max_blocks, int64_t *failed_at) {
    const int64_t bsz = s->buffer_size;
    int64_t blocks = 0;
    double *buf = (double *)malloc(sizeof(double) * (size_t)bsz);
    *failed_at = -1;
    for (int64_t t = 0; t < len; t++) {
        s->ring[s->write_pos] = data[t];
        s->write_pos = (s->write_pos + 1) % s->ring_len;
        s->in_buffer++;
        if (s->in_buffer >= bsz) {
            int64_t read_pos = (s->write_pos - bsz + s->ring_len) % s->ring_len;
            int finite = 1;
            for (int64_t i = 0; i < bsz; i++) {
                buf[i] = s->ring[(read_pos + i) % s->ring_len];
                if (!isfinite(buf[i])) finite = 0;
            }
            if (!finite) {
                *failed_at = t;
                break;
            }
            if (blocks < max_blocks)
                vwo_modwt_forward(buf, bsz, h, lh, g, lg, boundary, out + (blocks * 2 + 1) * bsz,
                                  out + (blocks * 2) * bsz);
            blocks++;
            s->in_buffer -= bsz - s->overlap;
        }
    }
    free(buf);
    return blocks;
}

/* :205-231 (flush; close does the same, :255-280): the unconsumed samples, zero-padded to bufferSize,
 * make one last block; returns 1 if a block was written to out[2][bufferSize], else 0 */
int vwo_stream_xform_flush(vwo_stream_xform *s, const double *h, int lh, const double *g, int lg,
                           int boundary, double *out) {
    if (s->in_buffer <= 0) return 0;
    const int64_t bsz = s->buffer_size;
    double *buf = (double *)calloc((size_t)bsz, sizeof(double));
    int64_t read_pos = (s->write_pos - s->in_buffer + s->ring_len) % s->ring_len;
    for (int64_t i = 0; i < s->in_buffer; i++) buf[i] = s->ring[(read_pos + i) % s->ring_len];
    vwo_modwt_forward(buf, bsz, h, lh, g, lg, boundary, out + bsz, out);
    free(buf);
    s->in_buffer = 0;
    s->write_pos = 0;
    return 1;
}

/* modwt/streaming/MultiLevelMODWTStreamingTransform.java:112-196: non-overlapping blocks of bufferSize
 * samples, each decomposed to `levels` levels; with flush != 0 the unconsumed samples, zero-padded, make a last
 * block (:166-196).  The whole stream is given at once and walked sample by sample through the ring buffer of
 * the reference.  Blocks go to out[block][levels+1][bufferSize] (d_1..d_J, a_J; the reference publishes the
 * levels of a block one by one, the approximation with the last).  Returns the number of blocks, or -status. */
int64_t vwo_stream_multilevel(const double *data, int64_t len, int64_t buffer_size, int levels, const double *h,
                              int lh, const double *g, int lg, int boundary, int flush, double *out,
                              int64_t max_blocks) {
    double *ring = (double *)calloc((size_t)buffer_size, sizeof(double));
    double *buf = (double *)malloc(sizeof(double) * (size_t)buffer_size);
    int64_t pos = 0, count = 0, blocks = 0;
    int status = 0;
    for (int64_t t = 0; t < len && status == 0; t++) {
        ring[pos] = data[t];
        pos = (pos + 1) % buffer_size;
        count++;
        if (count >= buffer_size) {
            int64_t start = (pos - buffer_size + buffer_size) % buffer_size;
            for (int64_t i = 0; i < buffer_size; i++) buf[i] = ring[(start + i) % buffer_size];
            if (blocks < max_blocks)
                status = vwo_multilevel_decompose(buf, buffer_size, levels, h, lh, g, lg, boundary,
                                                  out + blocks * (levels + 1) * buffer_size);
            blocks++;
            count = 0;
        }
    }
    if (flush && count > 0 && status == 0) {
        memset(buf, 0, sizeof(double) * (size_t)buffer_size);
        int64_t start = (pos - count + buffer_size) % buffer_size;
        for (int64_t i = 0; i < count; i++) buf[i] = ring[(start + i) % buffer_size];
        if (blocks < max_blocks)
            status = vwo_multilevel_decompose(buf, buffer_size, levels, h, lh, g, lg, boundary,
                                              out + blocks * (levels + 1) * buffer_size);
        blocks++;
    }
    free(ring);
    free(buf);
    return status != 0 ? -(int64_t)status : blocks;
}

/* ------------------------------------------------------------------------ */
/* swt/SWTResult.java: sparse detail bands */
/* ------------------------------------------------------------------------ */

/* SWTResult.SparseVector(dense, threshold) (swt/SWTResult.java:267-290): the entries with
 * Math.abs(v) > threshold in index order; returns their number (indices/values may be NULL to count). */
int64_t vwo_sparse_vector(const double *dense, int64_t n, double threshold, int32_t *indices, double *values) {
    int64_t count = 0;
    for (int64_t i = 0; i < n; i++) {
        if (fabs(dense[i]) > threshold) {
            if (indices) indices[count] = (int32_t)i;
            if (values) values[count] = dense[i];
            count++;
        }
    }
    return count;
}

/* SparseVector.toDense (:292-298) */
void vwo_sparse_to_dense(const int32_t *indices, const double *values, int64_t count, int64_t n, double *dense) {
    for (int64_t i = 0; i < n; i++) dense[i] = 0.0;
    for (int64_t i = 0; i < count; i++) dense[indices[i]] = values[i];
}

/* ------------------------------------------------------------------------ */
/* modwt/BatchSIMDMODWT.java (structure-of-arrays batch) and ParallelMultiLevelMODWT.java */
/* ------------------------------------------------------------------------ */

/* modwt/BatchSIMDMODWT.java:282-291: soa[t * batch + b] = signals[b][t] */
void vwo_convert_to_soa(const double *aos, int64_t batch, int64_t n, double *soa) {
    for (int64_t t = 0; t < n; t++)
        for (int64_t b = 0; b < batch; b++) soa[t * batch + b] = aos[b * n + t];
}

/* modwt/BatchSIMDMODWT.java:299-308 */
void vwo_convert_from_soa(const double *soa, int64_t batch, int64_t n, double *aos) {
    for (int64_t t = 0; t < n; t++)
        for (int64_t b = 0; b < batch; b++) aos[b * n + t] = soa[t * batch + b];
}

/* modwt/BatchSIMDMODWT.java:64-81 (dispatch): Haar -> :86-139 (constants 0.5, no running sum),
 * four taps -> :144-211 (taps * 1/sqrt(2), products chained without a zero start),
 * else -> :216-274 (running sums from zero).  Lane-wise vector arithmetic = the same scalar
 * operations per (t, b).  Source index (t - l + N) % N.  `hi` must hold at least flen taps (the
 * reference indexes highPass[l] for l < lowPass.length and throws when it is shorter). */
void vwo_batch_modwt_soa(const double *soa, double *soa_approx, double *soa_detail, const double *lo,
                         const double *hi, int flen, int is_haar, int64_t batch, int64_t n) {
    const double scale = 1.0 / sqrt(2.0);
    if (is_haar) {
        const double h0 = 0.5, h1 = 0.5, g0 = 0.5, g1 = -0.5;
        for (int64_t t = 0; t < n; t++) {
            int64_t tm1 = (t - 1 + n) % n;
            for (int64_t b = 0; b < batch; b++) {
                double s0 = soa[t * batch + b], s1 = soa[tm1 * batch + b];
                soa_approx[t * batch + b] = s0 * h0 + s1 * h1;
                soa_detail[t * batch + b] = s0 * g0 + s1 * g1;
            }
        }
    } else if (flen == 4) {
        double h[4], g[4];
        for (int i = 0; i < 4; i++) {
            h[i] = lo[i] * scale;
            g[i] = hi[i] * scale;
        }
        for (int64_t t = 0; t < n; t++) {
            int64_t i1 = (t - 1 + n) % n, i2 = (t - 2 + n) % n, i3 = (t - 3 + n) % n;
            for (int64_t b = 0; b < batch; b++) {
                double s0 = soa[t * batch + b], s1 = soa[i1 * batch + b], s2 = soa[i2 * batch + b],
                       s3 = soa[i3 * batch + b];
                soa_approx[t * batch + b] = ((s0 * h[0] + s1 * h[1]) + s2 * h[2]) + s3 * h[3];
                soa_detail[t * batch + b] = ((s0 * g[0] + s1 * g[1]) + s2 * g[2]) + s3 * g[3];
            }
        }
    } else {
        double *h = scaled_copy(lo, flen), *g = scaled_copy(hi, flen);
        for (int64_t t = 0; t < n; t++) {
            for (int64_t b = 0; b < batch; b++) {
                double sa = 0.0, sd = 0.0;
                for (int l = 0; l < flen; l++) {
                    int64_t src = (t - l + n) % n;
                    double v = soa[src * batch + b];
                    sa = sa + v * h[l];
                    sd = sd + v * g[l];
                }
                soa_approx[t * batch + b] = sa;
                soa_detail[t * batch + b] = sd;
            }
        }
        free(h);
        free(g);
    }
}

/* The batch-vectorised scheme of modwt/BatchSIMDMODWT.java:216-274 (general path: running sums from
 * zero, source index (t - l + N) % N, vector lanes along the batch of a structure-of-arrays block
 * soa[t * batch + b]) applied level by level with the level's dilation 2^(j-1), i.e. the cascade of
 * modwt/MultiLevelMODWTTransform.java:251-265 carried out in the reference's SIMD batch layout, followed by
 * the inverse cascade of modwt/MultiLevelMODWTTransform.java:638-648 (pair sum lo*a + hi*d, taps ascending)
 * in the same layout.  PERIODIC only, like the reference's batch class.  The lanes perform the same scalar
 * operations in the same order as the per-signal path, so the results equal vwo_multilevel_decompose /
 * vwo_multilevel_reconstruct bit for bit; only the loop nest (t outer, taps, batch inner and vectorised by
 * the compiler; OpenMP over t) differs.  Timed by the benchmark as the second CPU baseline.
 * coeffs: (levels + 1) planes of n * batch doubles, plane j-1 = d_j, plane `levels` = a_J (running
 * approximation ping-pongs between that plane and `scratch`, n * batch doubles). */
void vwo_soa_multilevel_decompose(const double *soa, int64_t batch, int64_t n, int levels, const double *h,
                                  int lh, const double *g, int lg, double *coeffs, double *scratch) {
    double *lo = scaled_copy(h, lh), *hi = scaled_copy(g, lg);
    const int64_t plane = n * batch;
    const double *cur = soa;
    for (int j = 1; j <= levels; j++) {
        const int64_t s = (int64_t)1 << (j - 1);
        /* the last level must land in plane `levels` */
        double *va = ((levels - j) % 2 == 0) ? coeffs + (int64_t)levels * plane : scratch;
        double *wd = coeffs + (int64_t)(j - 1) * plane;
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < n; t++) {
            double *oa = va + t * batch, *od = wd + t * batch;
            for (int64_t b = 0; b < batch; b++) oa[b] = 0.0, od[b] = 0.0;
            const int lmax = lh > lg ? lh : lg;
            for (int l = 0; l < lmax; l++) {
                int64_t src = (t - (int64_t)l * s) % n;
                if (src < 0) src += n;
                const double *v = cur + src * batch;
                if (l < lh) {
                    const double f = lo[l];
                    for (int64_t b = 0; b < batch; b++) oa[b] = oa[b] + v[b] * f;
                }
                if (l < lg) {
                    const double f = hi[l];
                    for (int64_t b = 0; b < batch; b++) od[b] = od[b] + v[b] * f;
                }
            }
        }
        cur = va;
    }
    free(lo);
    free(hi);
}

void vwo_soa_multilevel_reconstruct(const double *coeffs, int64_t batch, int64_t n, int levels, const double *hr,
                                    int lhr, const double *gr, double *soa_out, double *scratch) {
    double *lo = scaled_copy(hr, lhr), *hi = scaled_copy(gr, lhr);
    const int64_t plane = n * batch;
    const double *cur = coeffs + (int64_t)levels * plane;
    for (int j = levels; j >= 1; j--) {
        const int64_t s = (int64_t)1 << (j - 1);
        double *out = ((j - 1) % 2 == 0) ? soa_out : scratch;
        const double *wd = coeffs + (int64_t)(j - 1) * plane;
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < n; t++) {
            double *o = out + t * batch;
            for (int64_t b = 0; b < batch; b++) o[b] = 0.0;
            for (int l = 0; l < lhr; l++) {
                const int64_t src = (t + (int64_t)l * s) % n;
                const double *a = cur + src * batch, *d = wd + src * batch;
                const double fl = lo[l], fh = hi[l];
                for (int64_t b = 0; b < batch; b++) o[b] = o[b] + (fl * a[b] + fh * d[b]);
            }
        }
        cur = out;
    }
    free(lo);
    free(hi);
}

/* modwt/ParallelMultiLevelMODWT.java:218-248 (calculateMaxLevels; note the extra
 * signalLength <= filterLength rule) */
int vwo_parallel_max_levels(int64_t n, int filter_len) {
    if (n <= filter_len) return 0;
    int max_level = 1;
    int64_t lm1 = filter_len - 1;
    while (max_level < 10) {
        if (max_level - 1 >= 31) break;
        int64_t scaled = lm1 * ((int64_t)1 << (max_level - 1)) + 1;
        if (scaled > n) break;
        max_level++;
    }
    return max_level - 1;
}

/* modwt/ParallelMultiLevelMODWT.java:59-152: levels in [1, max]; PERIODIC -> circular convolution with
 * the zero-stuffed, 1/sqrt(2)-scaled filters (:172-208), every other mode -> zero padding (:100-131).
 * Within its level limit no filter is longer than the signal, so the cascade equals
 * vwo_multilevel_decompose.  Returns 104 (CFG_INVALID_DECOMPOSITION_LEVEL-like) for bad levels. */
int vwo_parallel_decompose(const double *x, int64_t n, int levels, const double *h, int lh, const double *g,
                           int lg, int boundary, double *coeffs) {
    int maxl = vwo_parallel_max_levels(n, lh);
    if (levels < 1 || levels > maxl) return 104;
    return vwo_multilevel_decompose(x, n, levels, h, lh, g, lg, boundary == 0 ? 0 : 1, coeffs);
}

/* ------------------------------------------------------------------------ */
/* batch drivers (row loops, OpenMP over signals) used by tests + cpu_baseline */
/* ------------------------------------------------------------------------ */

/* modwt/MODWTTransform.java:537-587: `for b` loop, output [B][2][N] = (d_1, a_1) */
void vwo_modwt_forward_batch(const double *x, int64_t b, int64_t n, const double *h, int lh,
                             const double *g, int lg, int boundary, double *coeffs) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < b; i++)
        vwo_modwt_forward(x + i * n, n, h, lh, g, lg, boundary, coeffs + (i * 2 + 1) * n,
                          coeffs + (i * 2) * n);
}

/* modwt/MODWTTransform.java:592-662 */
void vwo_modwt_inverse_batch(const double *coeffs, int64_t b, int64_t n, const double *hr, int lhr,
                             const double *gr, int lgr, int boundary, int batch_variant,
                             double *x) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < b; i++)
        vwo_modwt_inverse(coeffs + (i * 2 + 1) * n, coeffs + (i * 2) * n, n, hr, lhr, gr, lgr,
                          boundary, batch_variant, x + i * n);
}

int vwo_multilevel_decompose_batch(const double *x, int64_t b, int64_t n, int levels,
                                   const double *h, int lh, const double *g, int lg, int boundary,
                                   double *coeffs) {
    int status = 0;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < b; i++) {
        int st = vwo_multilevel_decompose(x + i * n, n, levels, h, lh, g, lg, boundary,
                                          coeffs + i * (levels + 1) * n);
        if (st != 0) {
#pragma omp critical
            status = st;
        }
    }
    return status;
}

void vwo_multilevel_reconstruct_batch(const double *coeffs, int64_t b, int64_t n, int levels,
                                      const double *hr, int lhr, const double *gr, int lgr,
                                      int boundary, double *x) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < b; i++)
        vwo_multilevel_reconstruct(coeffs + i * (levels + 1) * n, n, levels, hr, lhr, gr, lgr,
                                   boundary, x + i * n);
}

int vwo_swt_denoise_batch(const double *x, int64_t b, int64_t n, int levels, const double *h,
                          int lh, const double *g, int lg, const double *hr, int lhr,
                          const double *gr, int lgr, int boundary, double threshold, int soft,
                          double *out) {
    int status = 0;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < b; i++) {
        int st = vwo_swt_denoise(x + i * n, n, levels, h, lh, g, lg, hr, lhr, gr, lgr, boundary,
                                 threshold, soft, out + i * n);
        if (st != 0) {
#pragma omp critical
            status = st;
        }
    }
    return status;
}

/* util/ValidationUtils.java:106-117: index of the first NaN/Inf, or -1 */
int64_t vwo_first_non_finite(const double *x, int64_t n) {
    for (int64_t i = 0; i < n; i++)
        if (!isfinite(x[i])) return i;
    return -1;
}
