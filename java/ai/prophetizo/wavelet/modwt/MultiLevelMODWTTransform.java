package ai.prophetizo.wavelet.modwt;

import ai.prophetizo.wavelet.api.BoundaryMode;
import ai.prophetizo.wavelet.api.Wavelet;
import ai.prophetizo.wavelet.exception.ErrorCode;
import ai.prophetizo.wavelet.exception.ErrorContext;
import ai.prophetizo.wavelet.exception.InvalidArgumentException;
import ai.prophetizo.wavelet.exception.InvalidSignalException;
import ai.prophetizo.wavelet.gpu.GpuModwtBackend;
import ai.prophetizo.wavelet.util.ValidationUtils;

import java.util.Objects;

/**
 * Drop-in for the reference's multi-level MODWT class (modwt/MultiLevelMODWTTransform.java:66): same package,
 * constructor, public methods, result containers and exception rules, with the à-trous cascade running on the GPU
 * through {@link GpuModwtBackend}.  A J-level decomposition or reconstruction is ONE downcall (the reference runs
 * 2J convolutions over freshly upsampled filters per signal, :251-265, :369-372, :528-553); the partial
 * reconstructions pass a band mask instead of building zero-filled copies (:404-407, :437-465).  Coefficients are
 * bit-identical to the reference's scalar path.
 *
 * <p>Added beside the reference's methods: {@link #decomposeBatch}, {@link #decomposeMutableBatch} and
 * {@link #reconstructBatch}, which move B equal-length signals in one downcall; that is where the GPU pays.
 */
public class MultiLevelMODWTTransform {

    /** :101 */
    private static final int MAX_DECOMPOSITION_LEVELS = 10;

    private final Wavelet wavelet;
    private final BoundaryMode boundaryMode;
    private final MODWTTransform singleLevelTransform;

    /** :186-190: the single-level constructor supplies the boundary-mode check. */
    public MultiLevelMODWTTransform(Wavelet wavelet, BoundaryMode boundaryMode) {
        this.wavelet = Objects.requireNonNull(wavelet, "wavelet cannot be null");
        this.boundaryMode = Objects.requireNonNull(boundaryMode, "boundaryMode cannot be null");
        this.singleLevelTransform = new MODWTTransform(wavelet, boundaryMode);
    }

    /** :477-523: largest J with (L - 1) * 2^(J-1) + 1 <= N, below the level cap; evaluated by vw_max_levels. */
    private int calculateMaxLevels(int signalLength) {
        return GpuModwtBackend.maxLevels(signalLength, wavelet.lowPassDecomposition().length);
    }

    /**
     * The reference's argument checks in the reference's order (:216-244): non-finite samples first (a null signal
     * is a NullPointerException there, as here), then the empty signal, then the level count.
     */
    private void validate(double[] signal, int levels, String transformType) {
        ValidationUtils.validateFiniteValues(signal, "signal");
        if (signal.length == 0) {
            throw new InvalidSignalException(ErrorCode.VAL_EMPTY,
                ErrorContext.builder("Signal cannot be empty for multi-level MODWT")
                    .withContext("Transform type", transformType)
                    .withWavelet(wavelet)
                    .withBoundaryMode(boundaryMode)
                    .withContext("Requested levels", levels)
                    .build());
        }
        int maxLevels = calculateMaxLevels(signal.length);
        if (levels < 1 || levels > maxLevels) {
            throw new InvalidArgumentException(ErrorCode.CFG_INVALID_DECOMPOSITION_LEVEL,
                ErrorContext.builder("Invalid number of decomposition levels")
                    .withLevelInfo(levels, maxLevels)
                    .withSignalInfo(signal.length)
                    .withWavelet(wavelet)
                    .withContext("Filter length", wavelet.lowPassDecomposition().length)
                    .withSuggestion("Choose a level between 1 and " + maxLevels)
                    .build());
        }
    }

    /** Band k of one signal's block [J+1][N] (d_1..d_J, a_J). */
    private static double[] band(double[] block, int k, int n) {
        double[] out = new double[n];
        System.arraycopy(block, k * n, out, 0, n);
        return out;
    }

    private static MultiLevelMODWTResult immutableOf(double[] block, int levels, int n) {
        MultiLevelMODWTResultImpl r = new MultiLevelMODWTResultImpl(n, levels);
        for (int level = 1; level <= levels; level++) {
            r.setDetailCoeffsAtLevel(level, band(block, level - 1, n));
        }
        r.setApproximationCoeffs(band(block, levels, n));
        return r;
    }

    private static MutableMultiLevelMODWTResult mutableOf(double[] block, int levels, int n) {
        MutableMultiLevelMODWTResultImpl r = new MutableMultiLevelMODWTResultImpl(n, levels);
        for (int level = 1; level <= levels; level++) {
            r.setDetailCoeffsAtLevelDirect(level, band(block, level - 1, n));
        }
        r.setApproximationCoeffsDirect(band(block, levels, n));
        return r;
    }

    /** One block [J+1][N] (d_1..d_J, a_J) per result; the results share levels and length. */
    private static double[][] blocksOf(MultiLevelMODWTResult[] results, int levels, int n) {
        double[][] blocks = new double[results.length][Math.multiplyExact(levels + 1, n)];
        for (int b = 0; b < results.length; b++) {
            MultiLevelMODWTResult r = Objects.requireNonNull(results[b], "result cannot be null");
            if (r.getLevels() != levels || r.getSignalLength() != n) {
                throw new InvalidArgumentException(ErrorCode.VAL_LENGTH_MISMATCH,
                    "Results of one batch must share the level count and the signal length");
            }
            for (int level = 1; level <= levels; level++) {
                System.arraycopy(r.getDetailCoeffsAtLevel(level), 0, blocks[b], (level - 1) * n, n);
            }
            System.arraycopy(r.getApproximationCoeffs(), 0, blocks[b], levels * n, n);
        }
        return blocks;
    }

    private double[][] decomposeBlocks(double[][] signals, int levels) {
        return GpuModwtBackend.shared().forward(signals, wavelet, boundaryMode, levels, false, false);
    }

    /** :200-203 */
    public MultiLevelMODWTResult decompose(double[] signal) {
        return decompose(signal, calculateMaxLevels(signal.length));
    }

    /** :214-268 */
    public MultiLevelMODWTResult decompose(double[] signal, int levels) {
        validate(signal, levels, "Multi-level MODWT");
        return immutableOf(decomposeBlocks(new double[][] {signal}, levels)[0], levels, signal.length);
    }

    /** :281-284 */
    public MutableMultiLevelMODWTResult decomposeMutable(double[] signal) {
        return decomposeMutable(signal, calculateMaxLevels(signal.length));
    }

    /** :299-353 */
    public MutableMultiLevelMODWTResult decomposeMutable(double[] signal, int levels) {
        validate(signal, levels, "Multi-level MODWT (mutable)");
        return mutableOf(decomposeBlocks(new double[][] {signal}, levels)[0], levels, signal.length);
    }

    /**
     * decompose(signal, levels) of every row of an equal-length batch in one downcall (not in the reference).  Each
     * row is validated like decompose() validates its signal.
     */
    public MultiLevelMODWTResult[] decomposeBatch(double[][] signals, int levels) {
        Objects.requireNonNull(signals, "signals array cannot be null");
        for (double[] s : signals) {
            validate(s, levels, "Multi-level MODWT");
        }
        MultiLevelMODWTResult[] out = new MultiLevelMODWTResult[signals.length];
        if (signals.length == 0) {
            return out;
        }
        double[][] blocks = decomposeBlocks(signals, levels);
        for (int b = 0; b < signals.length; b++) {
            out[b] = immutableOf(blocks[b], levels, signals[0].length);
        }
        return out;
    }

    /** decomposeMutable(signal, levels) of every row of an equal-length batch in one downcall (not in the reference). */
    public MutableMultiLevelMODWTResult[] decomposeMutableBatch(double[][] signals, int levels) {
        Objects.requireNonNull(signals, "signals array cannot be null");
        for (double[] s : signals) {
            validate(s, levels, "Multi-level MODWT (mutable)");
        }
        MutableMultiLevelMODWTResult[] out = new MutableMultiLevelMODWTResult[signals.length];
        if (signals.length == 0) {
            return out;
        }
        double[][] blocks = decomposeBlocks(signals, levels);
        for (int b = 0; b < signals.length; b++) {
            out[b] = mutableOf(blocks[b], levels, signals[0].length);
        }
        return out;
    }

    private double[] reconstructMasked(MultiLevelMODWTResult result, int bandMask) {
        int levels = result.getLevels();
        int n = result.getSignalLength();
        double[][] blocks = blocksOf(new MultiLevelMODWTResult[] {result}, levels, n);
        return GpuModwtBackend.shared().inverse(blocks, n, wavelet, boundaryMode, levels, false, false, bandMask)[0];
    }

    /** :362-375 */
    public double[] reconstruct(MultiLevelMODWTResult result) {
        Objects.requireNonNull(result, "result cannot be null");
        return reconstructMasked(result, -1);
    }

    /** reconstruct(result) of B results of equal level count and length in one downcall (not in the reference). */
    public double[][] reconstructBatch(MultiLevelMODWTResult[] results) {
        Objects.requireNonNull(results, "results array cannot be null");
        if (results.length == 0) {
            return new double[0][];
        }
        int levels = Objects.requireNonNull(results[0], "result cannot be null").getLevels();
        int n = results[0].getSignalLength();
        return GpuModwtBackend.shared().inverse(blocksOf(results, levels, n), n, wavelet, boundaryMode, levels, false, false, -1);
    }

    /**
     * :385-410: the approximation and the details of startLevel..J take part, finer details count as zeros.
     * The exception carries no error code, as in the reference (:389-391).
     */
    public double[] reconstructFromLevel(MultiLevelMODWTResult result, int startLevel) {
        Objects.requireNonNull(result, "result cannot be null");
        int levels = result.getLevels();
        if (startLevel < 1 || startLevel > levels) {
            throw new InvalidArgumentException(
                "Invalid start level: " + startLevel + ". Must be between 1 and " + levels);
        }
        int mask = 1 << levels;
        for (int level = startLevel; level <= levels; level++) {
            mask |= 1 << (level - 1);
        }
        return reconstructMasked(result, mask);
    }

    /**
     * :420-468: band-pass.  Details of minLevel..maxLevel take part; the approximation only when the coarsest
     * level lies inside the range (:456-460).
     */
    public double[] reconstructLevels(MultiLevelMODWTResult result, int minLevel, int maxLevel) {
        Objects.requireNonNull(result, "result cannot be null");
        int levels = result.getLevels();
        if (minLevel < 1 || maxLevel > levels || minLevel > maxLevel) {
            throw new InvalidArgumentException(ErrorCode.CFG_INVALID_DECOMPOSITION_LEVEL,
                ErrorContext.builder("Invalid level range for partial reconstruction")
                    .withContext("Requested range", "[" + minLevel + ", " + maxLevel + "]")
                    .withContext("Available levels", levels)
                    .withContext("Transform type", "Multi-level MODWT reconstruction")
                    .build());
        }
        int mask = maxLevel == levels ? 1 << levels : 0;
        for (int level = minLevel; level <= maxLevel; level++) {
            mask |= 1 << (level - 1);
        }
        return reconstructMasked(result, mask);
    }

    public Wavelet getWavelet() {
        return wavelet;
    }

    public BoundaryMode getBoundaryMode() {
        return boundaryMode;
    }

    /** :776-778 */
    public int getMaximumLevels(int signalLength) {
        return Math.min(calculateMaxLevels(signalLength), MAX_DECOMPOSITION_LEVELS);
    }

    /** :785-787 */
    public static int getMaxDecompositionLevels() {
        return MAX_DECOMPOSITION_LEVELS;
    }
}
