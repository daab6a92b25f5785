package ai.prophetizo.wavelet.modwt;

import ai.prophetizo.wavelet.api.DiscreteWavelet;
import ai.prophetizo.wavelet.gpu.GpuModwtBackend;

/**
 * Drop-in for the reference's structure-of-arrays batch MODWT (modwt/BatchSIMDMODWT.java:32): the same static
 * methods on the same sample-major layout, {@code soa[t * batchSize + b]}, with the transform on the GPU
 * (vw_batch_modwt_soa).  Single level, periodic boundary, as the reference's three kernels (:86-274).
 *
 * <p>The reference's Haar kernel computes (x[t] + x[t-1]) * (1/sqrt 2) where the general kernel multiplies each
 * sample by its tap (:86-140 against :211-274); the native side keeps both forms, chosen by the wavelet's class as
 * here, so that results are bit-identical either way.  The layout conversions are host loops as in the reference:
 * they move no arithmetic.  There is no CPU fallback for the transform.
 */
public final class BatchSIMDMODWT {

    private BatchSIMDMODWT() {
    }

    /**
     * :64-81.  Arrays shorter than batchSize * signalLength fail with ArrayIndexOutOfBoundsException as the
     * reference's indexing would, before anything is sent to the GPU.
     */
    public static void batchMODWTSoA(double[] soaSignals, double[] soaApprox, double[] soaDetail,
                                     DiscreteWavelet wavelet, int batchSize, int signalLength) {
        long count = (long) batchSize * signalLength;
        if (batchSize <= 0 || signalLength <= 0) {
            return;   // the reference's loops do not execute
        }
        if (count > soaSignals.length || count > soaApprox.length || count > soaDetail.length) {
            throw new ArrayIndexOutOfBoundsException("batchSize * signalLength = " + count + " exceeds an array of "
                + Math.min(soaSignals.length, Math.min(soaApprox.length, soaDetail.length)) + " values");
        }
        GpuModwtBackend.shared().batchMODWTSoA(soaSignals, soaApprox, soaDetail, wavelet, batchSize, signalLength);
    }

    /** :282-291 */
    public static void convertToSoA(double[][] signals, double[] soaOutput) {
        int batchSize = signals.length;
        int signalLength = signals[0].length;
        for (int b = 0; b < batchSize; b++) {
            double[] row = signals[b];
            for (int t = 0; t < signalLength; t++) {
                soaOutput[t * batchSize + b] = row[t];
            }
        }
    }

    /** :299-308 */
    public static void convertFromSoA(double[] soaData, double[][] output) {
        int batchSize = output.length;
        int signalLength = output[0].length;
        for (int b = 0; b < batchSize; b++) {
            double[] row = output[b];
            for (int t = 0; t < signalLength; t++) {
                row[t] = soaData[t * batchSize + b];
            }
        }
    }

    /**
     * :322-328.  The reference clears its thread-local work arrays afterwards; this class keeps none (staging
     * buffers belong to the backend and are pooled per device, not per thread).
     */
    public static void batchMODWTWithCleanup(double[] soaSignals, double[] soaApprox, double[] soaDetail,
                                             DiscreteWavelet wavelet, int batchSize, int signalLength) {
        batchMODWTSoA(soaSignals, soaApprox, soaDetail, wavelet, batchSize, signalLength);
    }
}
