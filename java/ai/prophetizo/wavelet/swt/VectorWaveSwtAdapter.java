package ai.prophetizo.wavelet.swt;

import ai.prophetizo.wavelet.api.BoundaryMode;
import ai.prophetizo.wavelet.api.Wavelet;
import ai.prophetizo.wavelet.gpu.GpuModwtBackend;
import ai.prophetizo.wavelet.modwt.MultiLevelMODWTTransform;
import ai.prophetizo.wavelet.modwt.MutableMultiLevelMODWTResult;

import java.util.Map;
import java.util.Objects;
import java.util.concurrent.ConcurrentHashMap;

/**
 * Drop-in for the reference's SWT adapter (swt/VectorWaveSwtAdapter.java:61): same package, constructors, public
 * methods and exception rules over the GPU-backed {@link MultiLevelMODWTTransform}.
 *
 * <p>denoise() is ONE downcall: decomposition, the exact median of |d_1| behind the universal threshold,
 * thresholding of d_1..d_J and reconstruction run in one pass on the GPU (short rows never leave the chip between
 * the steps), where the reference makes four passes over [J+1][N] coefficients on the host (:296-312).
 * extractLevel() reconstructs through a band mask instead of zero-filling the other bands (:330-347).  The
 * reference's filter cache and thread pool have nothing left to do; cleanup() and getCacheStatistics() keep their
 * signatures and report that.  {@link #denoiseBatch} (not in the reference) denoises B equal-length signals at once.
 */
public class VectorWaveSwtAdapter {

    private static final int PARALLEL_THRESHOLD = 4096;  // :68, reported by getCacheStatistics()

    private final MultiLevelMODWTTransform modwtTransform;
    private final Wavelet wavelet;
    private final BoundaryMode boundaryMode;

    /** :122-131 */
    public VectorWaveSwtAdapter(Wavelet wavelet, BoundaryMode boundaryMode) {
        this.wavelet = Objects.requireNonNull(wavelet, "Wavelet cannot be null");
        this.boundaryMode = Objects.requireNonNull(boundaryMode, "Boundary mode cannot be null");
        this.modwtTransform = new MultiLevelMODWTTransform(wavelet, boundaryMode);
    }

    /** :150-152 */
    public VectorWaveSwtAdapter(Wavelet wavelet) {
        this(wavelet, BoundaryMode.PERIODIC);
    }

    /** :161-163 */
    public MutableMultiLevelMODWTResult forward(double[] signal) {
        return modwtTransform.decomposeMutable(signal);
    }

    /** :173-179 (the reference's "parallel" branch calls the same method, :185-196) */
    public MutableMultiLevelMODWTResult forward(double[] signal, int levels) {
        return modwtTransform.decomposeMutable(signal, levels);
    }

    /** :222-224 */
    public double[] inverse(MutableMultiLevelMODWTResult result) {
        return modwtTransform.reconstruct(result);
    }

    /** :239-243: the container's own rule (modwt/MutableMultiLevelMODWTResult.java:83-114); level 0 = approximation. */
    public void applyThreshold(MutableMultiLevelMODWTResult result, int level, double threshold, boolean soft) {
        Objects.requireNonNull(result, "Result cannot be null");
        result.applyThreshold(level, threshold, soft);
    }

    /**
     * :255-270.  sigma = median(|d_1|) / 0.6745 from the exact order statistics of the reference's sort (:377-395),
     * threshold = sigma * sqrt(2 ln N); the median is selected on the GPU, sqrt and log are the JVM's.
     */
    public void applyUniversalThreshold(MutableMultiLevelMODWTResult result, boolean soft) {
        Objects.requireNonNull(result, "Result cannot be null");
        int levels = result.getLevels();
        int n = result.getSignalLength();
        double[] block = new double[(levels + 1) * n];
        System.arraycopy(result.getMutableDetailCoeffs(1), 0, block, 0, n);  // only d_1 is read
        double threshold = GpuModwtBackend.shared().universalThresholds(new double[][] {block}, n, levels)[0];
        for (int level = 1; level <= levels; level++) {
            applyThreshold(result, level, threshold, soft);
        }
    }

    /** :282-284 */
    public double[] denoise(double[] signal, int levels) {
        return denoise(signal, levels, -1, true);
    }

    /**
     * :296-312.  forward() validates first in the reference, so the same checks run here before the fused call:
     * null, non-finite, empty, level count, in that order (modwt/MultiLevelMODWTTransform.java:301-329).
     */
    public double[] denoise(double[] signal, int levels, double threshold, boolean soft) {
        return denoiseBatch(new double[][] {signal}, levels, threshold, soft)[0];
    }

    /** denoise(signal, levels, threshold, soft) of every row of an equal-length batch in one downcall. */
    public double[][] denoiseBatch(double[][] signals, int levels, double threshold, boolean soft) {
        Objects.requireNonNull(signals, "signals array cannot be null");
        for (double[] s : signals) {
            validateLikeForward(s, levels);
        }
        return GpuModwtBackend.shared().denoise(signals, wavelet, boundaryMode, levels, threshold, soft);
    }

    private void validateLikeForward(double[] signal, int levels) {
        ai.prophetizo.wavelet.util.ValidationUtils.validateFiniteValues(signal, "signal");
        if (signal.length == 0 || levels < 1 || levels > GpuModwtBackend.maxLevels(signal.length,
            wavelet.lowPassDecomposition().length)) {
            // decomposeMutable rejects exactly these cases, with the reference's exception and error code
            modwtTransform.decomposeMutable(signal, levels);
            throw new IllegalStateException("unreachable: decomposeMutable accepted an invalid request");
        }
    }

    /**
     * :326-348: the band of targetLevel (0 = approximation) alone, reconstructed.  A targetLevel outside 0..levels
     * keeps nothing, as the reference's zeroing loops do.
     */
    public double[] extractLevel(double[] signal, int levels, int targetLevel) {
        MutableMultiLevelMODWTResult result = forward(signal, levels);
        if (targetLevel == 0) {
            // details finer than J+1 do not exist: only the approximation takes part
            return reconstructOnly(result, levels, 1 << levels);
        }
        return reconstructOnly(result, levels, targetLevel >= 1 && targetLevel <= levels ? 1 << (targetLevel - 1) : 0);
    }

    private double[] reconstructOnly(MutableMultiLevelMODWTResult result, int levels, int mask) {
        int n = result.getSignalLength();
        double[] block = new double[(levels + 1) * n];
        for (int level = 1; level <= levels; level++) {
            System.arraycopy(result.getMutableDetailCoeffs(level), 0, block, (level - 1) * n, n);
        }
        System.arraycopy(result.getMutableApproximationCoeffs(), 0, block, levels * n, n);
        return GpuModwtBackend.shared().inverse(new double[][] {block}, n, wavelet, boundaryMode, levels, false, false, mask)[0];
    }

    public Wavelet getWavelet() {
        return wavelet;
    }

    public BoundaryMode getBoundaryMode() {
        return boundaryMode;
    }

    /** :402-415: nothing is cached on the host any more. */
    public void cleanup() {
    }

    /** :423-429: same keys; the cache is empty and no executor exists. */
    public Map<String, Object> getCacheStatistics() {
        Map<String, Object> stats = new ConcurrentHashMap<>();
        stats.put("filterCacheSize", 0);
        stats.put("parallelExecutorActive", false);
        stats.put("parallelThreshold", PARALLEL_THRESHOLD);
        return stats;
    }
}
