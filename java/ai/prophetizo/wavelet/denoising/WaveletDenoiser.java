package ai.prophetizo.wavelet.denoising;

import ai.prophetizo.wavelet.api.BoundaryMode;
import ai.prophetizo.wavelet.api.Wavelet;
import ai.prophetizo.wavelet.exception.ErrorCode;
import ai.prophetizo.wavelet.exception.ErrorContext;
import ai.prophetizo.wavelet.exception.InvalidArgumentException;
import ai.prophetizo.wavelet.exception.InvalidSignalException;
import ai.prophetizo.wavelet.gpu.GpuModwtBackend;
import ai.prophetizo.wavelet.modwt.MultiLevelMODWTTransform;
import ai.prophetizo.wavelet.util.ValidationUtils;

import java.util.Objects;

/**
 * Drop-in for the reference's denoiser (denoising/WaveletDenoiser.java:43) with everything between the noisy and
 * the denoised signal on the GPU: forward MODWT, the MAD noise estimate of the finest detail band (:369-380), the
 * threshold rule (:385-427), thresholding (:544-568) and the inverse are one downcall per batch
 * (vw_wavelet_denoise_batch).  Same package, constructor, public methods, nested enums and exception rules.
 *
 * <p>Bit-identical to the reference for UNIVERSAL, MINIMAX, BAYES and fixed thresholds; SURE within the tolerance
 * the native header states for it.  The {@code *Batch} methods are additions: B equal-length signals per downcall.
 * There is no CPU fallback: without a usable GPU the first call throws IllegalStateException.
 */
public class WaveletDenoiser {

    /** :56: 1 &lt;&lt; (level - 1) must not overflow an int. */
    private static final int MAX_SAFE_LEVEL_FOR_SCALING = 31;

    private final Wavelet wavelet;
    private final BoundaryMode boundaryMode;

    /** :79-90 */
    public WaveletDenoiser(Wavelet wavelet, BoundaryMode boundaryMode) {
        if (wavelet == null) {
            throw InvalidArgumentException.nullArgument("wavelet");
        }
        if (boundaryMode == null) {
            throw InvalidArgumentException.nullArgument("boundaryMode");
        }
        this.wavelet = wavelet;
        this.boundaryMode = boundaryMode;
    }

    /** :96-98: DB4 with periodic boundaries. */
    public static WaveletDenoiser forFinancialData() {
        return new WaveletDenoiser(ai.prophetizo.wavelet.api.Daubechies.DB4, BoundaryMode.PERIODIC);
    }

    /** :108-110 */
    public double[] denoise(double[] signal, ThresholdMethod method) {
        return denoise(signal, method, ThresholdType.SOFT);
    }

    /**
     * :121-139.  The reference validates inside MODWTTransform.forward (boundary mode, then null, empty, non-finite)
     * and only afterwards meets the method: FIXED has no automatic rule (:405-415).
     */
    public double[] denoise(double[] signal, ThresholdMethod method, ThresholdType type) {
        checkBoundaryMode();
        checkSignal(signal, "MODWT");
        Objects.requireNonNull(method, "method cannot be null");
        Objects.requireNonNull(type, "type cannot be null");
        if (method == ThresholdMethod.FIXED) {
            throw fixedNeedsValue("denoise");
        }
        return GpuModwtBackend.shared().waveletDenoise(new double[][] {signal}, wavelet, boundaryMode, 0,
            method.ordinal(), type == ThresholdType.SOFT, 0.0)[0];
    }

    /**
     * :152-167: J-level decomposition, sigma from level 1, threshold of level j computed with sigma / sqrt(2^(j-1))
     * (:221-222), reconstruction from the thresholded bands.  Level validation is MultiLevelMODWTTransform's.
     */
    public double[] denoiseMultiLevel(double[] signal, int levels, ThresholdMethod method, ThresholdType type) {
        MultiLevelMODWTTransform multiTransform = new MultiLevelMODWTTransform(wavelet, boundaryMode);
        checkSignal(signal, "Multi-level MODWT");
        int maxLevels = multiTransform.getMaximumLevels(signal.length);
        if (levels < 1 || levels > maxLevels) {
            throw new InvalidArgumentException(ErrorCode.CFG_INVALID_DECOMPOSITION_LEVEL,
                ErrorContext.builder("Invalid number of decomposition levels")
                    .withLevelInfo(levels, maxLevels)
                    .withWavelet(wavelet)
                    .withSignalInfo(signal.length)
                    .build());
        }
        if (levels > MAX_SAFE_LEVEL_FOR_SCALING) {   // :188-200
            throw new InvalidArgumentException(ErrorCode.VAL_TOO_LARGE,
                ErrorContext.builder("Decomposition level exceeds safe limit for scale-dependent thresholds")
                    .withLevelInfo(levels, MAX_SAFE_LEVEL_FOR_SCALING)
                    .build());
        }
        Objects.requireNonNull(method, "method cannot be null");
        Objects.requireNonNull(type, "type cannot be null");
        if (method == ThresholdMethod.FIXED) {
            throw fixedNeedsValue("denoiseMultiLevel");
        }
        return GpuModwtBackend.shared().waveletDenoise(new double[][] {signal}, wavelet, boundaryMode, levels,
            method.ordinal(), type == ThresholdType.SOFT, 0.0)[0];
    }

    /** :350-360 */
    public double[] denoiseFixed(double[] signal, double threshold, ThresholdType type) {
        checkBoundaryMode();
        checkSignal(signal, "MODWT");
        Objects.requireNonNull(type, "type cannot be null");
        return GpuModwtBackend.shared().waveletDenoise(new double[][] {signal}, wavelet, boundaryMode, 0,
            ThresholdMethod.FIXED.ordinal(), type == ThresholdType.SOFT, threshold)[0];
    }

    /**
     * denoise(signal, method, type) of every row of an equal-length batch in one downcall (not in the reference).
     * Batches larger than the staging cap are cut into row chunks by the backend.
     */
    public double[][] denoiseBatch(double[][] signals, ThresholdMethod method, ThresholdType type) {
        checkBoundaryMode();
        Objects.requireNonNull(signals, "signals array cannot be null");
        for (double[] s : signals) {
            checkSignal(s, "MODWT");
        }
        if (method == ThresholdMethod.FIXED) {
            throw fixedNeedsValue("denoiseBatch");
        }
        return GpuModwtBackend.shared().waveletDenoise(signals, wavelet, boundaryMode, 0, method.ordinal(),
            type == ThresholdType.SOFT, 0.0);
    }

    /** denoiseMultiLevel of every row of an equal-length batch in one downcall (not in the reference). */
    public double[][] denoiseMultiLevelBatch(double[][] signals, int levels, ThresholdMethod method,
                                             ThresholdType type) {
        MultiLevelMODWTTransform multiTransform = new MultiLevelMODWTTransform(wavelet, boundaryMode);
        Objects.requireNonNull(signals, "signals array cannot be null");
        for (double[] s : signals) {
            checkSignal(s, "Multi-level MODWT");
        }
        if (signals.length > 0) {
            int maxLevels = Math.min(multiTransform.getMaximumLevels(signals[0].length), MAX_SAFE_LEVEL_FOR_SCALING);
            if (levels < 1 || levels > maxLevels) {
                throw new InvalidArgumentException(ErrorCode.CFG_INVALID_DECOMPOSITION_LEVEL,
                    ErrorContext.builder("Invalid number of decomposition levels")
                        .withLevelInfo(levels, maxLevels)
                        .withWavelet(wavelet)
                        .withSignalInfo(signals[0].length)
                        .build());
            }
        }
        if (method == ThresholdMethod.FIXED) {
            throw fixedNeedsValue("denoiseMultiLevelBatch");
        }
        return GpuModwtBackend.shared().waveletDenoise(signals, wavelet, boundaryMode, levels, method.ordinal(),
            type == ThresholdType.SOFT, 0.0);
    }

    /** What `new MODWTTransform(wavelet, boundaryMode)` rejects at the top of denoise / denoiseFixed. */
    private void checkBoundaryMode() {
        if (boundaryMode != BoundaryMode.PERIODIC && boundaryMode != BoundaryMode.ZERO_PADDING
            && boundaryMode != BoundaryMode.SYMMETRIC) {
            throw new InvalidArgumentException(ErrorCode.CFG_UNSUPPORTED_BOUNDARY_MODE,
                ErrorContext.builder("MODWT only supports PERIODIC, ZERO_PADDING, and SYMMETRIC boundary modes")
                    .withBoundaryMode(boundaryMode)
                    .withWavelet(wavelet)
                    .withContext("Transform type", "MODWT")
                    .build());
        }
    }

    /** The transforms' input checks, in their order: null, empty, non-finite (with the first bad index). */
    private void checkSignal(double[] signal, String transformType) {
        Objects.requireNonNull(signal, "signal cannot be null");
        if (signal.length == 0) {
            throw new InvalidSignalException(ErrorCode.VAL_EMPTY,
                ErrorContext.builder("Signal cannot be empty")
                    .withContext("Transform type", transformType)
                    .withWavelet(wavelet)
                    .withBoundaryMode(boundaryMode)
                    .build());
        }
        ValidationUtils.validateFiniteValues(signal, "signal");
    }

    private static InvalidArgumentException fixedNeedsValue(String operation) {
        return new InvalidArgumentException(ErrorCode.CFG_UNSUPPORTED_OPERATION,
            ErrorContext.builder("Fixed threshold method requires explicit threshold value")
                .withContext("Operation", operation)
                .withContext("Threshold method", "FIXED")
                .withSuggestion("Use denoiseFixed() method with explicit threshold value")
                .build());
    }

    /** :588-617; the ordinals are the native VW_THRESHOLD_* values. */
    public enum ThresholdMethod {
        /** sigma * sqrt(2 ln n) */
        UNIVERSAL,
        /** minimum of Stein's unbiased risk estimate */
        SURE,
        /** minimax risk threshold */
        MINIMAX,
        /** sigma^2 / sigma_x */
        BAYES,
        /** caller's value (denoiseFixed) */
        FIXED
    }

    /** :622-634 */
    public enum ThresholdType {
        /** shrink towards zero by the threshold */
        SOFT,
        /** keep or zero */
        HARD
    }
}
