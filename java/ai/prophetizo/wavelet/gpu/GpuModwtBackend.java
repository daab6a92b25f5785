package ai.prophetizo.wavelet.gpu;

import ai.prophetizo.wavelet.api.BoundaryMode;
import ai.prophetizo.wavelet.api.Wavelet;
import ai.prophetizo.wavelet.exception.ErrorCode;
import ai.prophetizo.wavelet.exception.InvalidArgumentException;
import ai.prophetizo.wavelet.exception.InvalidSignalException;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;
import java.util.ArrayDeque;
import java.util.concurrent.ConcurrentHashMap;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_DOUBLE;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

/**
 * The whole-batch seam that replaces the per-signal WaveletOperations / ScalarOps calls inside
 * MODWTTransform, MultiLevelMODWTTransform and VectorWaveSwtAdapter.  Java {@code double[][]} batches are
 * packed into one contiguous off-heap segment [B][N]; coefficients come back as [B][J+1][N] ordered
 * d_1..d_J, a_J.  One instance per GPU; safe for concurrent use (the native context serialises).
 *
 * <p>Signal and coefficient staging buffers are page-locked (vw_host_alloc) and reused from a small pool, so
 * that the copies to and from the GPU run by DMA at the PCIe rate; only the few filter taps and scalar
 * outputs come from a confined Arena.
 */
public final class GpuModwtBackend implements AutoCloseable {
    private static final ConcurrentHashMap<Integer, GpuModwtBackend> SHARED = new ConcurrentHashMap<>();

    private final MemorySegment ctx;
    private final ArrayDeque<MemorySegment> freeStaging = new ArrayDeque<>();
    private volatile boolean closed;

    public GpuModwtBackend(int device) {
        this.ctx = VwGpuNative.init(device);
    }

    /**
     * The process-wide backend of the device named by the system property {@code vectorwave.gpu.device}
     * (default 0); what the drop-in transform classes use.  Throws IllegalStateException when there is no
     * usable GPU: there is no CPU fallback behind this class.
     */
    public static GpuModwtBackend shared() {
        int device = Integer.getInteger("vectorwave.gpu.device", 0);
        return SHARED.computeIfAbsent(device, GpuModwtBackend::new);
    }

    /** vw_performance_info of this backend's context. */
    public String performanceInfo() {
        return VwGpuNative.performanceInfo(ctx);
    }

    /** MultiLevelMODWTTransform.calculateMaxLevels (modwt/MultiLevelMODWTTransform.java:477-523). */
    public static int maxLevels(int signalLength, int filterLength) {
        try {
            return (int) VwGpuNative.VW_MAX_LEVELS.invokeExact((long) signalLength, filterLength);
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    // ---- staging -------------------------------------------------------------------------------------

    /** A page-locked segment of at least {@code bytes} bytes; give it back with {@link #release}. */
    private MemorySegment acquire(long bytes) {
        synchronized (freeStaging) {
            for (var it = freeStaging.iterator(); it.hasNext(); ) {
                MemorySegment s = it.next();
                if (s.byteSize() >= bytes) {
                    it.remove();
                    return s;
                }
            }
        }
        // sizes rounded up to a power of two (at least 64 KiB) so that repeated calls of similar size share buffers
        long cap = Math.max(1L << 16, Long.highestOneBit(Math.max(bytes, 1) - 1) << 1);
        return VwGpuNative.hostAlloc(ctx, cap);
    }

    private void release(MemorySegment s) {
        synchronized (freeStaging) {
            if (!closed && freeStaging.size() < 8) {
                freeStaging.addFirst(s);
                return;
            }
        }
        VwGpuNative.hostFree(ctx, s);
    }

    private static void pack(MemorySegment seg, double[][] rows, int n) {
        for (int b = 0; b < rows.length; b++) {
            if (rows[b].length != n) {
                throw new InvalidArgumentException(ErrorCode.VAL_LENGTH_MISMATCH,
                    "Row " + b + " has " + rows[b].length + " samples, the batch has " + n);
            }
            MemorySegment.copy(rows[b], 0, seg, JAVA_DOUBLE, (long) b * n * Double.BYTES, n);
        }
    }

    private static void unpack(MemorySegment seg, double[][] rows, int n) {
        for (int b = 0; b < rows.length; b++) {
            MemorySegment.copy(seg, JAVA_DOUBLE, (long) b * n * Double.BYTES, rows[b], 0, n);
        }
    }

    private static MemorySegment taps(Arena a, double[] t) {
        return a.allocateFrom(JAVA_DOUBLE, t);
    }

    static int boundary(BoundaryMode m) {
        return switch (m) {
            case PERIODIC -> VwGpuNative.BOUNDARY_PERIODIC;
            case ZERO_PADDING -> VwGpuNative.BOUNDARY_ZERO_PADDING;
            case SYMMETRIC -> VwGpuNative.BOUNDARY_SYMMETRIC;
            default -> throw new InvalidArgumentException(ErrorCode.CFG_UNSUPPORTED_BOUNDARY_MODE,
                "MODWT only supports PERIODIC, ZERO_PADDING, and SYMMETRIC boundary modes");
        };
    }

    /** Maps a vw_status onto the exception class the reference's CPU path throws for the same condition. */
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
            case VwGpuNative.VW_E_NOMEM -> throw new OutOfMemoryError(msg);
            default -> throw new IllegalStateException("libvwgpu status " + status + ": " + msg);
        }
    }

    // ---- transforms ----------------------------------------------------------------------------------

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
        long xBytes = (long) b * n * Double.BYTES;
        MemorySegment stage = acquire(xBytes + (long) out.length * Double.BYTES);
        try (Arena a = Arena.ofConfined()) {
            MemorySegment x = stage.asSlice(0, xBytes);
            MemorySegment c = stage.asSlice(xBytes, (long) out.length * Double.BYTES);
            pack(x, signals, n);
            MemorySegment bad = a.allocate(JAVA_LONG);
            double[] h = w.lowPassDecomposition();
            double[] g = w.highPassDecomposition();
            int flags = (singleLevel ? VwGpuNative.FLAG_SINGLE_LEVEL : 0) | (checkFinite ? VwGpuNative.FLAG_CHECK_FINITE : 0);
            int st = (int) VwGpuNative.VW_MODWT_FORWARD_BATCH.invokeExact(ctx, x, (long) b, (long) n, levels, taps(a, h),
                h.length, taps(a, g), g.length, boundary(mode), flags, c, bad);
            check(st, bad.get(JAVA_LONG, 0));
            MemorySegment.copy(c, JAVA_DOUBLE, 0, out, 0, out.length);
            return out;
        } catch (RuntimeException | Error e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        } finally {
            release(stage);
        }
    }

    /**
     * Inverse of {@link #forward}: coefficients [B][J+1][N] -> signals [B][N].  bandMask == -1 uses every band;
     * otherwise band k of the block counts as zeros unless bit k is set (bit 0 = d_1 .. bit J-1 = d_J, bit J = a_J):
     * reconstructFromLevel / reconstructLevels / extractLevel without zeroed copies.
     */
    public double[][] inverse(double[] coeffs, int b, int n, Wavelet w, BoundaryMode mode, int levels,
                              boolean singleLevel, boolean batchVariant, int bandMask) {
        double[][] out = new double[b][n];
        if (b == 0) {
            return out;
        }
        long cBytes = (long) coeffs.length * Double.BYTES;
        long xBytes = (long) b * n * Double.BYTES;
        MemorySegment stage = acquire(cBytes + xBytes);
        try (Arena a = Arena.ofConfined()) {
            MemorySegment c = stage.asSlice(0, cBytes);
            MemorySegment x = stage.asSlice(cBytes, xBytes);
            MemorySegment.copy(coeffs, 0, c, JAVA_DOUBLE, 0, coeffs.length);
            double[] hr = w.lowPassReconstruction();
            double[] gr = w.highPassReconstruction();
            int st;
            long firstBad = -1;
            if (bandMask == -1) {
                MemorySegment bad = a.allocate(JAVA_LONG);
                int flags = (singleLevel ? VwGpuNative.FLAG_SINGLE_LEVEL : 0)
                    | (batchVariant ? VwGpuNative.FLAG_SYMMETRIC_BATCH_VARIANT : 0);
                st = (int) VwGpuNative.VW_MODWT_INVERSE_BATCH.invokeExact(ctx, c, (long) b, (long) n, levels, taps(a, hr),
                    hr.length, taps(a, gr), gr.length, boundary(mode), flags, x, bad);
                firstBad = bad.get(JAVA_LONG, 0);
            } else {
                st = (int) VwGpuNative.VW_MODWT_INVERSE_BANDS_BATCH.invokeExact(ctx, c, (long) b, (long) n, levels,
                    taps(a, hr), hr.length, taps(a, gr), gr.length, boundary(mode), bandMask, x);
            }
            check(st, firstBad);
            unpack(x, out, n);
            return out;
        } catch (RuntimeException | Error e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        } finally {
            release(stage);
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
        long xBytes = (long) b * n * Double.BYTES;
        MemorySegment stage = acquire(2 * xBytes);
        try (Arena a = Arena.ofConfined()) {
            MemorySegment x = stage.asSlice(0, xBytes);
            MemorySegment y = stage.asSlice(xBytes, xBytes);
            pack(x, signals, n);
            double[] h = w.lowPassDecomposition();
            double[] g = w.highPassDecomposition();
            double[] hr = w.lowPassReconstruction();
            double[] gr = w.highPassReconstruction();
            // the transcendental stays on the JVM so the threshold matches Math.sqrt(2 * Math.log(n)) bit for bit
            double sqrt2lnN = Math.sqrt(2 * Math.log(n));
            int st = (int) VwGpuNative.VW_SWT_DENOISE_BATCH.invokeExact(ctx, x, (long) b, (long) n, levels, taps(a, h),
                h.length, taps(a, g), g.length, taps(a, hr), hr.length, taps(a, gr), gr.length, boundary(mode), threshold,
                sqrt2lnN, soft ? 1 : 0, y, MemorySegment.NULL);
            check(st, -1);
            unpack(y, out, n);
            return out;
        } catch (RuntimeException | Error e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        } finally {
            release(stage);
        }
    }

    /**
     * The thresholds VectorWaveSwtAdapter.applyUniversalThreshold derives (swt/VectorWaveSwtAdapter.java:255-264,
     * 377-395) for B coefficient blocks [B][J+1][N]: median(|d_1|) / 0.6745 * sqrt(2 ln N), the median from the
     * exact order statistics the reference's sort yields.
     */
    public double[] universalThresholds(double[] coeffs, int b, int n, int levels) {
        double[] out = new double[b];
        if (b == 0) {
            return out;
        }
        long cBytes = (long) coeffs.length * Double.BYTES;
        MemorySegment stage = acquire(cBytes + (long) b * Double.BYTES);
        try {
            MemorySegment c = stage.asSlice(0, cBytes);
            MemorySegment thr = stage.asSlice(cBytes, (long) b * Double.BYTES);
            MemorySegment.copy(coeffs, 0, c, JAVA_DOUBLE, 0, coeffs.length);
            int st = (int) VwGpuNative.VW_SWT_UNIVERSAL_THRESHOLD_BATCH.invokeExact(ctx, c, (long) b, (long) n, levels,
                Math.sqrt(2 * Math.log(n)), thr);
            check(st, -1);
            MemorySegment.copy(thr, JAVA_DOUBLE, 0, out, 0, b);
            return out;
        } catch (RuntimeException | Error e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        } finally {
            release(stage);
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
        long xBytes = (long) b * n * Double.BYTES;
        MemorySegment stage = acquire(2 * xBytes);
        try (Arena a = Arena.ofConfined()) {
            MemorySegment x = stage.asSlice(0, xBytes);
            MemorySegment y = stage.asSlice(xBytes, xBytes);
            pack(x, signals, n);
            double[] h = w.lowPassDecomposition();
            double[] g = w.highPassDecomposition();
            double[] hr = w.lowPassReconstruction();
            double[] gr = w.highPassReconstruction();
            int flags = levels == 0 ? VwGpuNative.FLAG_SINGLE_LEVEL : 0;
            // Math.log stays on the JVM so that the thresholds match the CPU path bit for bit
            int st = (int) VwGpuNative.VW_WAVELET_DENOISE_BATCH.invokeExact(ctx, x, (long) b, (long) n, Math.max(levels, 1),
                taps(a, h), h.length, taps(a, g), g.length, taps(a, hr), hr.length, taps(a, gr), gr.length,
                boundary(mode), flags, method, soft ? 1 : 0, fixedThreshold, Math.log(n), y, MemorySegment.NULL);
            check(st, -1);
            unpack(y, out, n);
            return out;
        } catch (RuntimeException | Error e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        } finally {
            release(stage);
        }
    }

    // ---- streaming denoiser ----------------------------------------------------------------------------

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
        } catch (RuntimeException | Error e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public void destroyStreamState(MemorySegment state) {
        try {
            int st = (int) VwGpuNative.VW_STREAM_STATE_DESTROY.invokeExact(ctx, state);
            check(st, -1);
        } catch (RuntimeException | Error e) {
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
        } catch (RuntimeException | Error e) {
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
        if (b == 0) {
            return out;
        }
        long xBytes = (long) b * n * Double.BYTES;
        MemorySegment stage = acquire(2 * xBytes);
        try (Arena a = Arena.ofConfined()) {
            MemorySegment x = stage.asSlice(0, xBytes);
            MemorySegment y = stage.asSlice(xBytes, xBytes);
            pack(x, blocks, n);
            double[] h = w.lowPassDecomposition();
            double[] g = w.highPassDecomposition();
            double[] hr = w.lowPassReconstruction();
            double[] gr = w.highPassReconstruction();
            // Math.log / Math.sqrt stay on the JVM so that the thresholds match the CPU path bit for bit
            double logN = Math.log(n);
            int st = (int) VwGpuNative.VW_STREAMING_DENOISE_BATCH.invokeExact(ctx, state, x, (long) b, (long) n, taps(a, h),
                h.length, taps(a, g), g.length, taps(a, hr), hr.length, taps(a, gr), gr.length, boundary(mode), 0, method,
                soft ? 1 : 0, thresholdMultiplier, noiseEstimation, logN, Math.sqrt(2.0 * logN), y, MemorySegment.NULL);
            check(st, -1);
            unpack(y, out, n);
            return out;
        } catch (RuntimeException | Error e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        } finally {
            release(stage);
        }
    }

    // ---- structure-of-arrays batch ---------------------------------------------------------------------

    /**
     * BatchSIMDMODWT.batchMODWTSoA: single-level PERIODIC MODWT of a sample-major batch, soa[t * batchSize + b].
     */
    public void batchMODWTSoA(double[] soaSignals, double[] soaApprox, double[] soaDetail, Wavelet w, int batchSize,
                              int signalLength) {
        int count = Math.multiplyExact(batchSize, signalLength);
        long bytes = (long) count * Double.BYTES;
        MemorySegment stage = acquire(3 * bytes);
        try (Arena a = Arena.ofConfined()) {
            MemorySegment in = stage.asSlice(0, bytes);
            MemorySegment ap = stage.asSlice(bytes, bytes);
            MemorySegment de = stage.asSlice(2 * bytes, bytes);
            MemorySegment.copy(soaSignals, 0, in, JAVA_DOUBLE, 0, count);
            double[] h = w.lowPassDecomposition();
            double[] g = w.highPassDecomposition();
            int st = (int) VwGpuNative.VW_BATCH_MODWT_SOA.invokeExact(ctx, in, (long) batchSize, (long) signalLength,
                taps(a, h), h.length, taps(a, g), g.length, w instanceof ai.prophetizo.wavelet.api.Haar ? 1 : 0, ap, de);
            check(st, -1);
            MemorySegment.copy(ap, JAVA_DOUBLE, 0, soaApprox, 0, count);
            MemorySegment.copy(de, JAVA_DOUBLE, 0, soaDetail, 0, count);
        } catch (RuntimeException | Error e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        } finally {
            release(stage);
        }
    }

    @Override
    public void close() {
        closed = true;
        synchronized (freeStaging) {
            while (!freeStaging.isEmpty()) {
                VwGpuNative.hostFree(ctx, freeStaging.removeFirst());
            }
        }
        try {
            int ignored = (int) VwGpuNative.VW_DESTROY.invokeExact(ctx);
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }
}
