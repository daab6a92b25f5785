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
import java.util.concurrent.atomic.AtomicBoolean;

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
 *
 * <p>Batches of any size: coefficients travel as one {@code double[]} of (J+1)*N values per signal, and a call is
 * cut into row chunks so that no staging request exceeds {@code vectorwave.gpu.stagingBytes} (system property,
 * default 1 GiB) and no flat index exceeds what a Java array or a 32-bit int can hold.  The 65536 x 16384, J = 6
 * batch (7.5e9 coefficients) is 57 downcalls of 1149 rows at the default.
 */
public final class GpuModwtBackend implements AutoCloseable {
    private static final ConcurrentHashMap<Integer, GpuModwtBackend> SHARED = new ConcurrentHashMap<>();

    private final MemorySegment ctx;
    private final int device;
    private final ArrayDeque<MemorySegment> freeStaging = new ArrayDeque<>();
    private final AtomicBoolean closed = new AtomicBoolean();
    /** staging segments handed out and not yet given back; the context is destroyed when the last one returns */
    private int outstanding;
    private boolean destroyed;

    public GpuModwtBackend(int device) {
        this.device = device;
        this.ctx = VwGpuNative.init(device);
    }

    /**
     * The process-wide backend of the device named by the system property {@code vectorwave.gpu.device}
     * (default 0); what the drop-in transform classes use.  Throws IllegalStateException when there is no
     * usable GPU: there is no CPU fallback behind this class.
     */
    public static GpuModwtBackend shared() {
        int device = Integer.getInteger("vectorwave.gpu.device", 0);
        // a backend that somebody closed is replaced, not handed out again
        return SHARED.compute(device, (d, cur) -> (cur == null || cur.closed.get()) ? new GpuModwtBackend(d) : cur);
    }

    /** Largest staging request of one downcall, in bytes (system property vectorwave.gpu.stagingBytes, default 1 GiB). */
    static long stagingCap() {
        return Math.max(1L << 16, Long.getLong("vectorwave.gpu.stagingBytes", 1L << 30));
    }

    /**
     * Rows per downcall for rows of {@code inPerRow} input and {@code outPerRow} output values: within the staging cap
     * and with every flat element index of a chunk below Integer.MAX_VALUE - 8 (the largest Java array).
     */
    static int rowsPerCall(long inPerRow, long outPerRow, int rows) {
        long perRowBytes = Math.max(1, (inPerRow + outPerRow) * Double.BYTES);
        long byBytes = Math.max(1, stagingCap() / perRowBytes);
        long byIndex = Math.max(1, (Integer.MAX_VALUE - 8L) / Math.max(1, Math.max(inPerRow, outPerRow)));
        return (int) Math.max(1, Math.min(rows, Math.min(byBytes, byIndex)));
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
            if (closed.get()) {
                throw new IllegalStateException("GpuModwtBackend of device " + device + " is closed");
            }
            outstanding++;
            for (var it = freeStaging.iterator(); it.hasNext(); ) {
                MemorySegment s = it.next();
                if (s.byteSize() >= bytes) {
                    it.remove();
                    return s;
                }
            }
        }
        try {
            // sizes rounded up to a power of two (at least 64 KiB) so that repeated calls of similar size share buffers
            long cap = Math.max(1L << 16, Long.highestOneBit(Math.max(bytes, 1) - 1) << 1);
            return VwGpuNative.hostAlloc(ctx, cap);
        } catch (RuntimeException | Error e) {
            synchronized (freeStaging) {
                outstanding--;
            }
            throw e;
        }
    }

    private void release(MemorySegment s) {
        boolean destroyNow;
        synchronized (freeStaging) {
            outstanding--;
            if (!closed.get() && freeStaging.size() < 8) {
                freeStaging.addFirst(s);
                return;
            }
            destroyNow = closed.get() && outstanding == 0 && !destroyed;
            if (destroyNow) {
                destroyed = true;
            }
        }
        VwGpuNative.hostFree(ctx, s);  // the context is still alive: it is destroyed only below, after the last segment
        if (destroyNow) {
            destroyContext();
        }
    }

    private void destroyContext() {
        try {
            int ignored = (int) VwGpuNative.VW_DESTROY.invokeExact(ctx);
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /** Rows [r0, r1) of a batch into a segment, row-major. */
    private static void pack(MemorySegment seg, double[][] rows, int r0, int r1, int n) {
        for (int b = r0; b < r1; b++) {
            if (rows[b].length != n) {
                throw new InvalidArgumentException(ErrorCode.VAL_LENGTH_MISMATCH,
                    "Row " + b + " has " + rows[b].length + " values, the batch has " + n);
            }
            MemorySegment.copy(rows[b], 0, seg, JAVA_DOUBLE, (long) (b - r0) * n * Double.BYTES, n);
        }
    }

    private static void unpack(MemorySegment seg, double[][] rows, int r0, int r1, int n) {
        for (int b = r0; b < r1; b++) {
            MemorySegment.copy(seg, JAVA_DOUBLE, (long) (b - r0) * n * Double.BYTES, rows[b], 0, n);
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
     * @return one block of (J+1)*N coefficients per signal, ordered d_1..d_J, a_J
     */
    public double[][] forward(double[][] signals, Wavelet w, BoundaryMode mode, int levels, boolean singleLevel,
                              boolean checkFinite) {
        int b = signals.length;
        int n = b == 0 ? 0 : signals[0].length;
        int perRow = Math.multiplyExact(levels + 1, n);  // one signal's block must fit a Java array, as in the reference
        double[][] out = new double[b][perRow];
        if (b == 0) {
            return out;
        }
        int step = rowsPerCall(n, perRow, b);
        double[] h = w.lowPassDecomposition();
        double[] g = w.highPassDecomposition();
        int flags = (singleLevel ? VwGpuNative.FLAG_SINGLE_LEVEL : 0) | (checkFinite ? VwGpuNative.FLAG_CHECK_FINITE : 0);
        MemorySegment stage = acquire((long) step * (n + (long) perRow) * Double.BYTES);
        try (Arena a = Arena.ofConfined()) {
            MemorySegment th = taps(a, h);
            MemorySegment tg = taps(a, g);
            MemorySegment bad = a.allocate(JAVA_LONG);
            for (int r0 = 0; r0 < b; r0 += step) {
                int r1 = Math.min(b, r0 + step);
                long xBytes = (long) (r1 - r0) * n * Double.BYTES;
                long cBytes = (long) (r1 - r0) * perRow * Double.BYTES;
                MemorySegment x = stage.asSlice(0, xBytes);
                MemorySegment c = stage.asSlice((long) step * n * Double.BYTES, cBytes);
                pack(x, signals, r0, r1, n);
                int st = (int) VwGpuNative.VW_MODWT_FORWARD_BATCH.invokeExact(ctx, x, (long) (r1 - r0), (long) n, levels, th,
                    h.length, tg, g.length, boundary(mode), flags, c, bad);
                long firstBad = bad.get(JAVA_LONG, 0);
                check(st, firstBad < 0 ? firstBad : firstBad + (long) r0 * n);  // flat index of the whole batch
                unpack(c, out, r0, r1, perRow);
            }
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
     * Inverse of {@link #forward}: one block of (J+1)*N coefficients per signal -> signals [B][N].  bandMask == -1 uses
     * every band; otherwise band k of the block counts as zeros unless bit k is set (bit 0 = d_1 .. bit J-1 = d_J,
     * bit J = a_J): reconstructFromLevel / reconstructLevels / extractLevel without zeroed copies.
     */
    public double[][] inverse(double[][] coeffs, int n, Wavelet w, BoundaryMode mode, int levels,
                              boolean singleLevel, boolean batchVariant, int bandMask) {
        int b = coeffs.length;
        int perRow = Math.multiplyExact(levels + 1, n);
        double[][] out = new double[b][n];
        if (b == 0) {
            return out;
        }
        int step = rowsPerCall(perRow, n, b);
        double[] hr = w.lowPassReconstruction();
        double[] gr = w.highPassReconstruction();
        MemorySegment stage = acquire((long) step * (n + (long) perRow) * Double.BYTES);
        try (Arena a = Arena.ofConfined()) {
            MemorySegment thr = taps(a, hr);
            MemorySegment tgr = taps(a, gr);
            MemorySegment bad = a.allocate(JAVA_LONG);
            for (int r0 = 0; r0 < b; r0 += step) {
                int r1 = Math.min(b, r0 + step);
                long cBytes = (long) (r1 - r0) * perRow * Double.BYTES;
                long xBytes = (long) (r1 - r0) * n * Double.BYTES;
                MemorySegment c = stage.asSlice(0, cBytes);
                MemorySegment x = stage.asSlice((long) step * perRow * Double.BYTES, xBytes);
                pack(c, coeffs, r0, r1, perRow);
                int st;
                long firstBad = -1;
                if (bandMask == -1) {
                    int flags = (singleLevel ? VwGpuNative.FLAG_SINGLE_LEVEL : 0)
                        | (batchVariant ? VwGpuNative.FLAG_SYMMETRIC_BATCH_VARIANT : 0);
                    st = (int) VwGpuNative.VW_MODWT_INVERSE_BATCH.invokeExact(ctx, c, (long) (r1 - r0), (long) n, levels, thr,
                        hr.length, tgr, gr.length, boundary(mode), flags, x, bad);
                    firstBad = bad.get(JAVA_LONG, 0);
                    if (firstBad >= 0) {
                        firstBad += (long) r0 * perRow;
                    }
                } else {
                    st = (int) VwGpuNative.VW_MODWT_INVERSE_BANDS_BATCH.invokeExact(ctx, c, (long) (r1 - r0), (long) n, levels,
                        thr, hr.length, tgr, gr.length, boundary(mode), bandMask, x);
                }
                check(st, firstBad);
                unpack(x, out, r0, r1, n);
            }
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
        int step = rowsPerCall(n, n, b);
        double[] h = w.lowPassDecomposition();
        double[] g = w.highPassDecomposition();
        double[] hr = w.lowPassReconstruction();
        double[] gr = w.highPassReconstruction();
        // the transcendental stays on the JVM so the threshold matches Math.sqrt(2 * Math.log(n)) bit for bit
        double sqrt2lnN = Math.sqrt(2 * Math.log(n));
        long half = (long) step * n * Double.BYTES;
        MemorySegment stage = acquire(2 * half);
        try (Arena a = Arena.ofConfined()) {
            MemorySegment th = taps(a, h);
            MemorySegment tg = taps(a, g);
            MemorySegment thr = taps(a, hr);
            MemorySegment tgr = taps(a, gr);
            for (int r0 = 0; r0 < b; r0 += step) {
                int r1 = Math.min(b, r0 + step);
                long bytes = (long) (r1 - r0) * n * Double.BYTES;
                MemorySegment x = stage.asSlice(0, bytes);
                MemorySegment y = stage.asSlice(half, bytes);
                pack(x, signals, r0, r1, n);
                int st = (int) VwGpuNative.VW_SWT_DENOISE_BATCH.invokeExact(ctx, x, (long) (r1 - r0), (long) n, levels, th,
                    h.length, tg, g.length, thr, hr.length, tgr, gr.length, boundary(mode), threshold,
                    sqrt2lnN, soft ? 1 : 0, y, MemorySegment.NULL);
                check(st, -1);
                unpack(y, out, r0, r1, n);
            }
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
    public double[] universalThresholds(double[][] coeffs, int n, int levels) {
        int b = coeffs.length;
        int perRow = Math.multiplyExact(levels + 1, n);
        double[] out = new double[b];
        if (b == 0) {
            return out;
        }
        int step = rowsPerCall(perRow, 1, b);
        long cCap = (long) step * perRow * Double.BYTES;
        MemorySegment stage = acquire(cCap + (long) step * Double.BYTES);
        try {
            for (int r0 = 0; r0 < b; r0 += step) {
                int r1 = Math.min(b, r0 + step);
                MemorySegment c = stage.asSlice(0, (long) (r1 - r0) * perRow * Double.BYTES);
                MemorySegment thr = stage.asSlice(cCap, (long) (r1 - r0) * Double.BYTES);
                pack(c, coeffs, r0, r1, perRow);
                int st = (int) VwGpuNative.VW_SWT_UNIVERSAL_THRESHOLD_BATCH.invokeExact(ctx, c, (long) (r1 - r0), (long) n, levels,
                    Math.sqrt(2 * Math.log(n)), thr);
                check(st, -1);
                MemorySegment.copy(thr, JAVA_DOUBLE, 0, out, r0, r1 - r0);
            }
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
        int step = rowsPerCall(n, n, b);
        double[] h = w.lowPassDecomposition();
        double[] g = w.highPassDecomposition();
        double[] hr = w.lowPassReconstruction();
        double[] gr = w.highPassReconstruction();
        int flags = levels == 0 ? VwGpuNative.FLAG_SINGLE_LEVEL : 0;
        long half = (long) step * n * Double.BYTES;
        MemorySegment stage = acquire(2 * half);
        try (Arena a = Arena.ofConfined()) {
            MemorySegment th = taps(a, h);
            MemorySegment tg = taps(a, g);
            MemorySegment thr = taps(a, hr);
            MemorySegment tgr = taps(a, gr);
            for (int r0 = 0; r0 < b; r0 += step) {
                int r1 = Math.min(b, r0 + step);
                long bytes = (long) (r1 - r0) * n * Double.BYTES;
                MemorySegment x = stage.asSlice(0, bytes);
                MemorySegment y = stage.asSlice(half, bytes);
                pack(x, signals, r0, r1, n);
                // Math.log stays on the JVM so that the thresholds match the CPU path bit for bit
                int st = (int) VwGpuNative.VW_WAVELET_DENOISE_BATCH.invokeExact(ctx, x, (long) (r1 - r0), (long) n,
                    Math.max(levels, 1), th, h.length, tg, g.length, thr, hr.length, tgr, gr.length,
                    boundary(mode), flags, method, soft ? 1 : 0, fixedThreshold, Math.log(n), y, MemorySegment.NULL);
                check(st, -1);
                unpack(y, out, r0, r1, n);
            }
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
        // (one block per stream and call: the state advances once per call, so this batch is not cut into chunks)
        long xBytes = Math.multiplyExact((long) b * n, (long) Double.BYTES);
        MemorySegment stage = acquire(2 * xBytes);
        try (Arena a = Arena.ofConfined()) {
            MemorySegment x = stage.asSlice(0, xBytes);
            MemorySegment y = stage.asSlice(xBytes, xBytes);
            pack(x, blocks, 0, b, n);
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
            unpack(y, out, 0, b, n);
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

    /**
     * Idempotent.  The first call wins; the instance leaves the shared map (so {@link #shared} builds a new one), pooled
     * staging buffers are freed, and the native context is destroyed now or, when calls are still in flight on other
     * threads, when the last of their staging segments comes back.
     */
    @Override
    public void close() {
        if (!closed.compareAndSet(false, true)) {
            return;
        }
        SHARED.remove(device, this);
        boolean destroyNow;
        synchronized (freeStaging) {
            while (!freeStaging.isEmpty()) {
                VwGpuNative.hostFree(ctx, freeStaging.removeFirst());
            }
            destroyNow = outstanding == 0 && !destroyed;
            if (destroyNow) {
                destroyed = true;
            }
        }
        if (destroyNow) {
            destroyContext();
        }
    }
}
