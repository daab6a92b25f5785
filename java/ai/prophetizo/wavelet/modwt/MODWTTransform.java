package ai.prophetizo.wavelet.modwt;

import ai.prophetizo.wavelet.WaveletOperations;
import ai.prophetizo.wavelet.api.BoundaryMode;
import ai.prophetizo.wavelet.api.Wavelet;
import ai.prophetizo.wavelet.exception.ErrorCode;
import ai.prophetizo.wavelet.exception.ErrorContext;
import ai.prophetizo.wavelet.exception.InvalidArgumentException;
import ai.prophetizo.wavelet.exception.InvalidSignalException;
import ai.prophetizo.wavelet.gpu.GpuModwtBackend;
import ai.prophetizo.wavelet.performance.AdaptivePerformanceEstimator;
import ai.prophetizo.wavelet.performance.PredictionResult;
import ai.prophetizo.wavelet.util.ValidationUtils;

import java.util.Objects;

/**
 * Drop-in for the reference's single-level MODWT class (modwt/MODWTTransform.java:69) whose arithmetic runs on
 * the GPU through {@link GpuModwtBackend}: same package, constructor, public methods, result types and exception
 * rules; the result containers, wavelets, exceptions and estimator classes are the reference's own, unchanged.
 *
 * <p>What moves: forward / inverse and the two batch methods.  A batch of equal-length signals is ONE downcall
 * (the reference loops over signals, :564-577, :616-659); a single signal is a batch of one.  Coefficients are
 * bit-identical to the reference's scalar path (products and sums rounded separately, taps in ascending order).
 * There is no CPU fallback: without a usable GPU the first transform call throws IllegalStateException.
 */
public class MODWTTransform {

    private final Wavelet wavelet;
    private final BoundaryMode boundaryMode;

    /** :83-103: null arguments are NullPointerExceptions, other boundary modes CFG_UNSUPPORTED_BOUNDARY_MODE. */
    public MODWTTransform(Wavelet wavelet, BoundaryMode boundaryMode) {
        this.wavelet = Objects.requireNonNull(wavelet, "wavelet cannot be null");
        this.boundaryMode = Objects.requireNonNull(boundaryMode, "boundaryMode cannot be null");
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

    /**
     * :123-179.  Validation order of the reference: null (NullPointerException), empty (VAL_EMPTY), non-finite
     * (ValidationUtils: the exception names the first bad index); it runs on the host, where the array is.
     */
    public MODWTResult forward(double[] signal) {
        Objects.requireNonNull(signal, "signal cannot be null");
        if (signal.length == 0) {
            throw new InvalidSignalException(ErrorCode.VAL_EMPTY,
                ErrorContext.builder("Signal cannot be empty")
                    .withContext("Transform type", "MODWT")
                    .withWavelet(wavelet)
                    .withBoundaryMode(boundaryMode)
                    .build());
        }
        ValidationUtils.validateFiniteValues(signal, "signal");
        double[][] c = GpuModwtBackend.shared().forward(new double[][] {signal}, wavelet, boundaryMode, 1, true, false);
        return resultOf(c[0], signal.length);
    }

    /** One signal's single-level coefficient block [2][N] (detail, approximation) as a MODWTResult. */
    private static MODWTResult resultOf(double[] block, int n) {
        double[] detail = new double[n];
        double[] approx = new double[n];
        System.arraycopy(block, 0, detail, 0, n);
        System.arraycopy(block, n, approx, 0, n);
        return MODWTResult.create(approx, detail);
    }

    /** One block [2][N] (detail, approximation) per result. */
    private static double[][] blocksOf(MODWTResult[] results, int n) {
        double[][] blocks = new double[results.length][Math.multiplyExact(2, n)];
        for (int b = 0; b < results.length; b++) {
            System.arraycopy(results[b].detailCoeffs(), 0, blocks[b], 0, n);
            System.arraycopy(results[b].approximationCoeffs(), 0, blocks[b], n, n);
        }
        return blocks;
    }

    /** :193-291: null -> NullPointerException; non-finite coefficients -> VAL_NON_FINITE_VALUES. */
    public double[] inverse(MODWTResult modwtResult) {
        Objects.requireNonNull(modwtResult, "modwtResult cannot be null");
        if (!modwtResult.isValid()) {
            throw new InvalidSignalException(ErrorCode.VAL_NON_FINITE_VALUES,
                ErrorContext.builder("MODWTResult contains invalid coefficients")
                    .withContext("Coefficient validity", "Contains NaN or Infinity values")
                    .withWavelet(wavelet)
                    .withBoundaryMode(boundaryMode)
                    .build());
        }
        int n = modwtResult.getSignalLength();
        double[][] blocks = blocksOf(new MODWTResult[] {modwtResult}, n);
        return GpuModwtBackend.shared().inverse(blocks, n, wavelet, boundaryMode, 1, true, false, -1)[0];
    }

    public Wavelet getWavelet() {
        return wavelet;
    }

    public BoundaryMode getBoundaryMode() {
        return boundaryMode;
    }

    /** :317-319.  The record describes the host's vector unit; {@link #getBackendInfo} describes the GPU path. */
    public WaveletOperations.PerformanceInfo getPerformanceInfo() {
        return WaveletOperations.getPerformanceInfo();
    }

    /** Description of the GPU back end the transforms run on (vw_performance_info); not in the reference. */
    public String getBackendInfo() {
        return GpuModwtBackend.shared().performanceInfo();
    }

    /** :328-348, unchanged: the adaptive estimator's model of the host path. */
    public ProcessingEstimate estimateProcessingTime(int signalLength) {
        var perfInfo = getPerformanceInfo();
        PredictionResult prediction = AdaptivePerformanceEstimator.getInstance()
            .estimateMODWT(signalLength, wavelet.name(), perfInfo.vectorizationEnabled());
        return new ProcessingEstimate(signalLength, prediction.estimatedTime(), perfInfo.vectorizationEnabled(),
            perfInfo.estimateSpeedup(signalLength), prediction.confidence(), prediction.lowerBound(),
            prediction.upperBound());
    }

    /** :404-442 */
    public record ProcessingEstimate(int signalLength, double estimatedTimeMs, boolean vectorizationUsed,
                                     double speedupFactor, double confidence, double lowerBoundMs,
                                     double upperBoundMs) {
        public String description() {
            String base = vectorizationUsed
                ? String.format("Signal length %d: %.2fms [%.2f-%.2fms] (%.1fx speedup with vectors)", signalLength,
                    estimatedTimeMs, lowerBoundMs, upperBoundMs, speedupFactor)
                : String.format("Signal length %d: %.2fms [%.2f-%.2fms] (scalar mode)", signalLength, estimatedTimeMs,
                    lowerBoundMs, upperBoundMs);
            return base + String.format(" - %.0f%% confidence", confidence * 100);
        }

        public boolean isFastProcessing() {
            return estimatedTimeMs < 1.0;
        }

        public boolean hasSignificantSpeedup() {
            return speedupFactor > 2.0;
        }
    }

    /**
     * :459-487.  Four or more signals of one length >= 64 take the reference's "optimized" route, which does not
     * validate its input (:537-587): a non-finite sample surfaces as the InvalidSignalException of
     * MODWTResult.create for the coefficients it poisons.  Everything else goes signal by signal through
     * {@link #forward}, with its validation.  Equal-length signals are one downcall either way round.
     */
    public MODWTResult[] forwardBatch(double[][] signals) {
        Objects.requireNonNull(signals, "signals array cannot be null");
        if (signals.length == 0) {
            return new MODWTResult[0];
        }
        int n = signals[0].length;
        boolean sameLengths = true;
        for (int i = 1; i < signals.length && sameLengths; i++) {
            sameLengths = signals[i].length == n;
        }
        MODWTResult[] results = new MODWTResult[signals.length];
        if (sameLengths && signals.length >= 4 && n >= 64) {
            double[][] c = GpuModwtBackend.shared().forward(signals, wavelet, boundaryMode, 1, true, false);
            for (int b = 0; b < signals.length; b++) {
                results[b] = resultOf(c[b], n);
            }
            return results;
        }
        for (int i = 0; i < signals.length; i++) {
            results[i] = forward(signals[i]);
        }
        return results;
    }

    /**
     * :504-532.  The optimized route (:592-662) skips isValid() and, for SYMMETRIC, reflects t + l where
     * inverse() reflects t - l (:644-657 against :269-287); the backend's batchVariant flag selects that form.
     */
    public double[][] inverseBatch(MODWTResult[] results) {
        Objects.requireNonNull(results, "results array cannot be null");
        if (results.length == 0) {
            return new double[0][];
        }
        int n = results[0].getSignalLength();
        boolean sameLengths = true;
        for (int i = 1; i < results.length && sameLengths; i++) {
            sameLengths = results[i].getSignalLength() == n;
        }
        if (sameLengths && results.length >= 4 && n >= 64) {
            return GpuModwtBackend.shared().inverse(blocksOf(results, n), n, wavelet, boundaryMode, 1, true, true, -1);
        }
        double[][] reconstructed = new double[results.length][];
        for (int i = 0; i < results.length; i++) {
            reconstructed[i] = inverse(results[i]);
        }
        return reconstructed;
    }
}
