package ai.prophetizo.wavelet.gpu;

import ai.prophetizo.wavelet.api.BoundaryMode;
import ai.prophetizo.wavelet.api.Wavelet;
import ai.prophetizo.wavelet.exception.ErrorCode;
import ai.prophetizo.wavelet.exception.InvalidArgumentException;
import ai.prophetizo.wavelet.exception.InvalidSignalException;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_DOUBLE;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

/**
 * The whole-batch seam that replaces the per-signal WaveletOperations / ScalarOps calls inside
 * MODWTTransform, MultiLevelMODWTTransform and VectorWaveSwtAdapter.  Java {@code double[][]} batches are
 * packed into one contiguous off-heap segment [B][N]; coefficients come back as [B][J+1][N] ordered
 * d_1..d_J, a_J.  One instance per GPU; safe for concurrent use (the native context serialises).
 */
public final class GpuModwtBackend implements AutoCloseable {
    private final MemorySegment ctx;

    public GpuModwtBackend(int device) {
        this.ctx = VwGpuNative.init(device);
    }

    private static int boundary(BoundaryMode m) {
        return switch (m) {
            case PERIODIC -> 0;
            case ZERO_PADDING -> 1;
            case SYMMETRIC -> 2;
            default -> throw new InvalidArgumentException(ErrorCode.CFG_UNSUPPORTED_BOUNDARY_MODE,
                "MODWT only supports PERIODIC, ZERO_PADDING, and SYMMETRIC boundary modes");
        };
    }

    private static MemorySegment pack(Arena a, double[][] rows, int n) {
        MemorySegment seg = a.allocate(JAVA_DOUBLE, (long) rows.length * n);
        for (int b = 0; b < rows.length; b++) {
            MemorySegment.copy(rows[b], 0, seg, JAVA_DOUBLE, (long) b * n * Double.BYTES, n);
        }
        return seg;
    }

    private static MemorySegment taps(Arena a, double[] t) {
        return a.allocateFrom(JAVA_DOUBLE, t);
    }

    private void check(int status, long firstBad) {
        if (status == VwGpuNative.VW_OK) {
            return;
        }
        String msg = VwGpuNative.lastError(ctx);
        switch (status) {
            case VwGpuNative.VW_E_NONFINITE -> throw new InvalidSignalException(ErrorCode.VAL_NON_FINITE_VALUES,
                msg + " (flat index " + firstBad + ")");
            case VwGpuNative.VW_E_EMPTY -> throw new InvalidSignalException(ErrorCode.VAL_EMPTY, msg);
            case VwGpuNative.VW_E_LEVEL -> throw new InvalidArgumentException(ErrorCode.CFG_INVALID_DECOMPOSITION_LEVEL, msg);
            case VwGpuNative.VW_E_BOUNDARY -> throw new InvalidArgumentException(ErrorCode.CFG_UNSUPPORTED_BOUNDARY_MODE, msg);
            case VwGpuNative.VW_E_TOO_LARGE -> throw new InvalidArgumentException(ErrorCode.VAL_TOO_LARGE, msg);
            case VwGpuNative.VW_E_UNSUPPORTED -> throw new InvalidArgumentException(ErrorCode.CFG_UNSUPPORTED_OPERATION, msg);
            case VwGpuNative.VW_E_NULL -> throw new NullPointerException(msg);
            case VwGpuNative.VW_E_NOT_POSITIVE -> throw new IllegalArgumentException(msg);
            case VwGpuNative.VW_E_LENGTH_MISMATCH -> throw new ArrayIndexOutOfBoundsException(msg);
            default -> throw new IllegalStateException("libvwgpu status " + status + ": " + msg);
        }
    }

    /**
     * J-level forward of B equal-length signals.  levels == 1 with singleLevel == true is
     * MODWTTransform.forwardBatch; otherwise MultiLevelMODWTTransform.decompose applied per row.
     *
     * @return coefficients [B][J+1][N] flattened
     */
    public double[] forward(double[][] signals, Wavelet w, BoundaryMode mode, int levels, boolean singleLevel,
                            boolean checkFinite) {
        int b = signals.length;
        int n = b == 0 ? 0 : signals[0].length;
        double[] out = new double[Math.multiplyExact(Math.multiplyExact(b, levels + 1), n)];
        if (b == 0) {
            return out;
        }
        try (Arena a = Arena.ofConfined()) {
            MemorySegment x = pack(a, signals, n);
            MemorySegment c = a.allocate(JAVA_DOUBLE, (long) out.length);
            MemorySegment bad = a.allocate(JAVA_LONG);
            double[] h = w.lowPassDecomposition();
            double[] g = w.highPassDecomposition();
            int flags = (singleLevel ? VwGpuNative.FLAG_SINGLE_LEVEL : 0) | (checkFinite ? VwGpuNative.FLAG_CHECK_FINITE : 0);
            int st = (int) VwGpuNative.VW_FORWARD.invokeExact(ctx, x, (long) b, (long) n, levels, taps(a, h), h.length,
                taps(a, g), g.length, boundary(mode), flags, c, bad);
            check(st, bad.get(JAVA_LONG, 0));
            MemorySegment.copy(c, JAVA_DOUBLE, 0, out, 0, out.length);
            return out;
        } catch (RuntimeException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /** Inverse of {@link #forward}: coefficients [B][J+1][N] -> signals [B][N]. */
    public double[][] inverse(double[] coeffs, int b, int n, Wavelet w, BoundaryMode mode, int levels,
                              boolean singleLevel, boolean batchVariant) {
        double[][] out = new double[b][n];
        if (b == 0) {
            return out;
        }
        try (Arena a = Arena.ofConfined()) {
            MemorySegment c = a.allocateFrom(JAVA_DOUBLE, coeffs);
            MemorySegment x = a.allocate(JAVA_DOUBLE, (long) b * n);
            MemorySegment bad = a.allocate(JAVA_LONG);
            double[] hr = w.lowPassReconstruction();
            double[] gr = w.highPassReconstruction();
            int flags = (singleLevel ? VwGpuNative.FLAG_SINGLE_LEVEL : 0)
                | (batchVariant ? VwGpuNative.FLAG_SYMMETRIC_BATCH_VARIANT : 0);
            int st = (int) VwGpuNative.VW_INVERSE.invokeExact(ctx, c, (long) b, (long) n, levels, taps(a, hr), hr.length,
                taps(a, gr), gr.length, boundary(mode), flags, x, bad);
            check(st, bad.get(JAVA_LONG, 0));
            for (int i = 0; i < b; i++) {
                MemorySegment.copy(x, JAVA_DOUBLE, (long) i * n * Double.BYTES, out[i], 0, n);
            }
            return out;
        } catch (RuntimeException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /**
     * WaveletDenoiser for a batch (denoising/WaveletDenoiser.java): levels == 0 is denoise()/denoiseFixed()
     * (single-level transform), levels &gt;= 1 is denoiseMultiLevel().  method: ThresholdMethod.ordinal()
     * (UNIVERSAL, SURE, MINIMAX, BAYES, FIXED).
     */
    public double[][] waveletDenoise(double[][] signals, Wavelet w, BoundaryMode mode, int levels, int method,
                                     boolean soft, double fixedThreshold) {
        int b = signals.length;
        int n = b == 0 ? 0 : signals[0].length;
        double[][] out = new double[b][n];
        if (b == 0) {
            return out;
        }
        try (Arena a = Arena.ofConfined()) {
            MemorySegment x = pack(a, signals, n);
            MemorySegment y = a.allocate(JAVA_DOUBLE, (long) b * n);
            double[] h = w.lowPassDecomposition();
            double[] g = w.highPassDecomposition();
            double[] hr = w.lowPassReconstruction();
            double[] gr = w.highPassReconstruction();
            int flags = levels == 0 ? 8 /* VW_FLAG_SINGLE_LEVEL */ : 0;
            // Math.log stays on the JVM so that the thresholds match the CPU path bit for bit
            int st = (int) VwGpuNative.VW_WAVELET_DENOISE.invokeExact(ctx, x, (long) b, (long) n, Math.max(levels, 1),
                taps(a, h), h.length, taps(a, g), g.length, taps(a, hr), hr.length, taps(a, gr), gr.length,
                boundary(mode), flags, method, soft ? 1 : 0, fixedThreshold, Math.log(n), y, MemorySegment.NULL);
            check(st, -1);
            for (int i = 0; i < b; i++) {
                MemorySegment.copy(y, JAVA_DOUBLE, (long) i * n * Double.BYTES, out[i], 0, n);
            }
            return out;
        } catch (RuntimeException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /**
     * State of {@code streams} lock-step MODWTStreamingDenoiser instances (noise rings and levels in GPU memory).
     * One per MODWTStreamingDenoiser object (streams == 1) or per channel group; release with
     * {@link #destroyStreamState}.
     */
    public MemorySegment createStreamState(long streams, int noiseWindowSize) {
        try (Arena a = Arena.ofConfined()) {
            MemorySegment out = a.allocate(ADDRESS);
            int st = (int) VwGpuNative.VW_STREAM_STATE_CREATE.invokeExact(ctx, streams, (long) noiseWindowSize, out);
            check(st, -1);
            return out.get(ADDRESS, 0);
        } catch (RuntimeException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public void destroyStreamState(MemorySegment state) {
        try {
            int st = (int) VwGpuNative.VW_STREAM_STATE_DESTROY.invokeExact(ctx, state);
            check(st, -1);
        } catch (RuntimeException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /** MODWTStreamingDenoiser.getEstimatedNoiseLevel() of every stream of the state. */
    public double[] estimatedNoiseLevels(MemorySegment state, int streams) {
        try (Arena a = Arena.ofConfined()) {
            MemorySegment lv = a.allocate(JAVA_DOUBLE, streams);
            int st = (int) VwGpuNative.VW_STREAM_STATE_NOISE_LEVELS.invokeExact(ctx, state, lv);
            check(st, -1);
            return lv.toArray(JAVA_DOUBLE);
        } catch (RuntimeException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /**
     * MODWTStreamingDenoiser.denoise(samples) for one block per stream of the state: blocks[streams][n].
     * method: ThresholdMethod.ordinal(); noiseEstimation: NoiseEstimation.ordinal() (MAD, STD, FIXED).
     */
    public double[][] streamingDenoise(MemorySegment state, double[][] blocks, Wavelet w, BoundaryMode mode, int method,
                                       boolean soft, double thresholdMultiplier, int noiseEstimation) {
        int b = blocks.length;
        int n = b == 0 ? 0 : blocks[0].length;
        double[][] out = new double[b][n];
        try (Arena a = Arena.ofConfined()) {
            MemorySegment x = pack(a, blocks, n);
            MemorySegment y = a.allocate(JAVA_DOUBLE, (long) b * n);
            double[] h = w.lowPassDecomposition();
            double[] g = w.highPassDecomposition();
            double[] hr = w.lowPassReconstruction();
            double[] gr = w.highPassReconstruction();
            // Math.log / Math.sqrt stay on the JVM so that the thresholds match the CPU path bit for bit
            double logN = Math.log(n);
            int st = (int) VwGpuNative.VW_STREAMING_DENOISE.invokeExact(ctx, state, x, (long) b, (long) n, taps(a, h), h.length,
                taps(a, g), g.length, taps(a, hr), hr.length, taps(a, gr), gr.length, boundary(mode), 0, method, soft ? 1 : 0,
                thresholdMultiplier, noiseEstimation, logN, Math.sqrt(2.0 * logN), y, MemorySegment.NULL);
            check(st, -1);
            for (int i = 0; i < b; i++) {
                MemorySegment.copy(y, JAVA_DOUBLE, (long) i * n * Double.BYTES, out[i], 0, n);
            }
            return out;
        } catch (RuntimeException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /**
     * BatchSIMDMODWT.batchMODWTSoA: single-level PERIODIC MODWT of a sample-major batch, soa[t * batchSize + b].
     */
    public void batchMODWTSoA(double[] soaSignals, double[] soaApprox, double[] soaDetail, Wavelet w, int batchSize,
                              int signalLength) {
        long count = (long) batchSize * signalLength;
        try (Arena a = Arena.ofConfined()) {
            MemorySegment in = a.allocate(JAVA_DOUBLE, count);
            MemorySegment.copy(soaSignals, 0, in, JAVA_DOUBLE, 0, (int) count);
            MemorySegment ap = a.allocate(JAVA_DOUBLE, count);
            MemorySegment de = a.allocate(JAVA_DOUBLE, count);
            double[] h = w.lowPassDecomposition();
            double[] g = w.highPassDecomposition();
            int st = (int) VwGpuNative.VW_BATCH_MODWT_SOA.invokeExact(ctx, in, (long) batchSize, (long) signalLength, taps(a, h),
                h.length, taps(a, g), g.length, w instanceof ai.prophetizo.wavelet.api.Haar ? 1 : 0, ap, de);
            check(st, -1);
            MemorySegment.copy(ap, JAVA_DOUBLE, 0, soaApprox, 0, (int) count);
            MemorySegment.copy(de, JAVA_DOUBLE, 0, soaDetail, 0, (int) count);
        } catch (RuntimeException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /** VectorWaveSwtAdapter.denoise for a batch; threshold &lt; 0 selects the universal threshold. */
    public double[][] denoise(double[][] signals, Wavelet w, BoundaryMode mode, int levels, double threshold,
                              boolean soft) {
        int b = signals.length;
        int n = b == 0 ? 0 : signals[0].length;
        double[][] out = new double[b][n];
        if (b == 0) {
            return out;
        }
        try (Arena a = Arena.ofConfined()) {
            MemorySegment x = pack(a, signals, n);
            MemorySegment y = a.allocate(JAVA_DOUBLE, (long) b * n);
            double[] h = w.lowPassDecomposition();
            double[] g = w.highPassDecomposition();
            double[] hr = w.lowPassReconstruction();
            double[] gr = w.highPassReconstruction();
            // the transcendental stays on the JVM so the threshold matches Math.sqrt(2 * Math.log(n)) bit for bit
            double sqrt2lnN = Math.sqrt(2 * Math.log(n));
            int st = (int) VwGpuNative.VW_DENOISE.invokeExact(ctx, x, (long) b, (long) n, levels, taps(a, h), h.length,
                taps(a, g), g.length, taps(a, hr), hr.length, taps(a, gr), gr.length, boundary(mode), threshold,
                sqrt2lnN, soft ? 1 : 0, y, MemorySegment.NULL);
            check(st, -1);
            for (int i = 0; i < b; i++) {
                MemorySegment.copy(y, JAVA_DOUBLE, (long) i * n * Double.BYTES, out[i], 0, n);
            }
            return out;
        } catch (RuntimeException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    @Override
    public void close() {
        try {
            int ignored = (int) VwGpuNative.VW_DESTROY.invokeExact(ctx);
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }
}
