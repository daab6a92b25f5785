package ai.prophetizo.wavelet.modwt;

import ai.prophetizo.wavelet.api.BoundaryMode;
import ai.prophetizo.wavelet.api.DiscreteWavelet;
import ai.prophetizo.wavelet.api.Wavelet;
import ai.prophetizo.wavelet.exception.InvalidArgumentException;
import ai.prophetizo.wavelet.exception.InvalidSignalException;
import ai.prophetizo.wavelet.gpu.GpuModwtBackend;
import ai.prophetizo.wavelet.util.ValidationUtils;

import java.util.concurrent.Executor;
import java.util.concurrent.ForkJoinPool;

/**
 * Drop-in for the reference's executor-driven cascade (modwt/ParallelMultiLevelMODWT.java:27): same package,
 * constructors, decompose signature, validation order and messages.  The reference runs the low-pass and the
 * high-pass convolution of each level as two tasks and chains the levels (:92-143); here all levels of the signal
 * run in one GPU launch, so the executor is kept only because callers pass one.
 *
 * <p>Arithmetic: within this class's level limit the zero-stuffed, 1/sqrt(2)^j-scaled filters (:173-210) never wrap
 * twice round the signal, so the coefficients are those of MultiLevelMODWTTransform.decompose, bit for bit.  Every
 * boundary mode other than PERIODIC is zero padding (:103-133).  {@link #decomposeBatch} is an addition.
 */
public class ParallelMultiLevelMODWT {

    private static final int MAX_DECOMPOSITION_LEVELS = 10;

    @SuppressWarnings("unused")
    private final Executor executor;

    /** :37-39 */
    public ParallelMultiLevelMODWT() {
        this(ForkJoinPool.commonPool());
    }

    /** :46-48 */
    public ParallelMultiLevelMODWT(Executor executor) {
        this.executor = executor;
    }

    /** :59-152.  Order of checks: finite values (null included), empty, wavelet kind, level range. */
    public MultiLevelMODWTResult decompose(double[] signal, Wavelet wavelet, BoundaryMode mode, int levels) {
        DiscreteWavelet dw = validate(signal, wavelet, levels);
        double[] block = GpuModwtBackend.shared()
            .forward(new double[][] {signal}, dw, effectiveMode(mode), levels, false, false)[0];
        return resultOf(block, levels, signal.length);
    }

    /**
     * decompose of every row of an equal-length batch in one downcall (not in the reference): the way to give the GPU
     * more than one signal's worth of work from this class.
     */
    public MultiLevelMODWTResult[] decomposeBatch(double[][] signals, Wavelet wavelet, BoundaryMode mode, int levels) {
        if (signals.length == 0) {
            return new MultiLevelMODWTResult[0];
        }
        DiscreteWavelet dw = null;
        for (double[] s : signals) {
            dw = validate(s, wavelet, levels);
        }
        double[][] blocks = GpuModwtBackend.shared().forward(signals, dw, effectiveMode(mode), levels, false, false);
        MultiLevelMODWTResult[] out = new MultiLevelMODWTResult[signals.length];
        for (int b = 0; b < signals.length; b++) {
            out[b] = resultOf(blocks[b], levels, signals[0].length);
        }
        return out;
    }

    private static BoundaryMode effectiveMode(BoundaryMode mode) {
        return mode == BoundaryMode.PERIODIC ? BoundaryMode.PERIODIC : BoundaryMode.ZERO_PADDING;
    }

    private static DiscreteWavelet validate(double[] signal, Wavelet wavelet, int levels) {
        ValidationUtils.validateFiniteValues(signal, "signal");
        if (signal.length == 0) {
            throw new InvalidSignalException("Signal cannot be empty");
        }
        if (!(wavelet instanceof DiscreteWavelet dw)) {
            throw new InvalidArgumentException("Multi-level MODWT requires a discrete wavelet");
        }
        int maxLevels = calculateMaxLevels(signal.length, dw);
        if (levels < 1 || levels > maxLevels) {
            throw new InvalidArgumentException(
                "Invalid number of levels: " + levels + ". Must be between 1 and " + maxLevels);
        }
        return dw;
    }

    private static MultiLevelMODWTResult resultOf(double[] block, int levels, int n) {
        MultiLevelMODWTResultImpl result = new MultiLevelMODWTResultImpl(n, levels);
        for (int level = 1; level <= levels; level++) {
            double[] d = new double[n];
            System.arraycopy(block, (level - 1) * n, d, 0, n);
            result.setDetailCoeffsAtLevel(level, d);
        }
        double[] a = new double[n];
        System.arraycopy(block, levels * n, a, 0, n);
        result.setApproximationCoeffs(a);
        return result;
    }

    /**
     * :225-257.  Unlike MultiLevelMODWTTransform's limit, a signal no longer than the filter has no level at all, and
     * the count stops below ten: the largest J &lt; 10 with (L - 1) * 2^(J-1) + 1 &lt;= N.
     */
    private static int calculateMaxLevels(int signalLength, DiscreteWavelet wavelet) {
        int filterLength = wavelet.lowPassDecomposition().length;
        if (signalLength <= filterLength) {
            return 0;
        }
        int level = 0;
        while (level + 1 < MAX_DECOMPOSITION_LEVELS
            && (long) (filterLength - 1) * (1L << level) + 1L <= signalLength) {
            level++;
        }
        return level;
    }
}
