package ai.prophetizo.wavelet.gpu;

import java.lang.foreign.Arena;
import java.lang.foreign.FunctionDescriptor;
import java.lang.foreign.Linker;
import java.lang.foreign.MemorySegment;
import java.lang.foreign.SymbolLookup;
import java.lang.invoke.MethodHandle;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_DOUBLE;
import static java.lang.foreign.ValueLayout.JAVA_INT;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

/**
 * FFM (java.lang.foreign) downcall handles for libvwgpu.so, the C ABI declared in include/vwgpu.h.
 * No JNI shim: every handle is a direct downcall over off-heap MemorySegments.  Requires
 * {@code --enable-native-access=ALL-UNNAMED} (or the module name) on Java 22+.
 *
 * <p>NOTE: written for Java 23 against the header; the build image of the GPU library has no JDK, so this
 * file has not been compiled there.
 */
public final class VwGpuNative {
    private static final Linker LINKER = Linker.nativeLinker();
    private static final SymbolLookup LIB =
        SymbolLookup.libraryLookup(System.getProperty("vectorwave.gpu.library", "libvwgpu.so"), Arena.global());

    // status codes (vw_status)
    public static final int VW_OK = 0, VW_E_NULL = 1, VW_E_NONFINITE = 3, VW_E_NOT_POSITIVE = 4, VW_E_TOO_LARGE = 5,
        VW_E_EMPTY = 6, VW_E_LENGTH_MISMATCH = 7, VW_E_BOUNDARY = 103, VW_E_LEVEL = 104, VW_E_UNSUPPORTED = 105, VW_E_NO_DEVICE = 900;
    // flags
    public static final int FLAG_CHECK_FINITE = 1, FLAG_SYMMETRIC_BATCH_VARIANT = 2, FLAG_NO_LEVEL_CHECK = 4,
        FLAG_SINGLE_LEVEL = 8, FLAG_FMA = 16;

    private static MethodHandle h(String name, FunctionDescriptor fd) {
        return LINKER.downcallHandle(LIB.find(name).orElseThrow(() -> new UnsatisfiedLinkError(name)), fd);
    }

    static final MethodHandle VW_INIT = h("vw_init", FunctionDescriptor.of(JAVA_INT, JAVA_INT, ADDRESS));
    static final MethodHandle VW_DESTROY = h("vw_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
    static final MethodHandle VW_LAST_ERROR = h("vw_last_error", FunctionDescriptor.of(ADDRESS, ADDRESS));
    static final MethodHandle VW_PERFORMANCE_INFO = h("vw_performance_info", FunctionDescriptor.of(ADDRESS, ADDRESS));
    static final MethodHandle VW_MAX_LEVELS = h("vw_max_levels", FunctionDescriptor.of(JAVA_INT, JAVA_LONG, JAVA_INT));

    // int32 vw_modwt_forward_batch(ctx, x, b, n, levels, h, lh, g, lg, boundary, flags, coeffs, first_bad)
    static final MethodHandle VW_FORWARD = h("vw_modwt_forward_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS));
    // int32 vw_modwt_inverse_batch(ctx, coeffs, b, n, levels, hr, lhr, gr, lgr, boundary, flags, x, first_bad)
    static final MethodHandle VW_INVERSE = h("vw_modwt_inverse_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS));
    // int32 vw_modwt_inverse_bands_batch(ctx, coeffs, b, n, levels, hr, lhr, gr, lgr, boundary, band_mask, x)
    static final MethodHandle VW_INVERSE_BANDS = h("vw_modwt_inverse_bands_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS));
    // int32 vw_swt_denoise_batch(ctx, x, b, n, levels, h, lh, g, lg, hr, lhr, gr, lgr, boundary, threshold,
    //                            sqrt_2_ln_n, soft, y, thresholds_out)
    static final MethodHandle VW_DENOISE = h("vw_swt_denoise_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_DOUBLE, JAVA_DOUBLE, JAVA_INT, ADDRESS,
            ADDRESS));
    // int32 vw_wavelet_denoise_batch(ctx, x, b, n, levels, h, lh, g, lg, hr, lhr, gr, lgr, boundary, flags, method,
    //                                soft, fixed_threshold, log_n, y, thresholds_out)
    static final MethodHandle VW_WAVELET_DENOISE = h("vw_wavelet_denoise_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_DOUBLE,
            JAVA_DOUBLE, ADDRESS, ADDRESS));
    // int32 vw_denoiser_thresholds_batch(ctx, coeffs, b, n, levels, method, level_scaled, log_n, thresholds)
    static final MethodHandle VW_DENOISER_THRESHOLDS = h("vw_denoiser_thresholds_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, JAVA_INT, JAVA_INT,
            JAVA_DOUBLE, ADDRESS));
    // noise estimation of the streaming denoiser (vw_noise_estimation)
    public static final int NOISE_MAD = 0, NOISE_STD = 1, NOISE_FIXED = 2;
    // int32 vw_stream_state_create(ctx, streams, window_size, out_state)
    static final MethodHandle VW_STREAM_STATE_CREATE = h("vw_stream_state_create",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS));
    static final MethodHandle VW_STREAM_STATE_DESTROY = h("vw_stream_state_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS));
    // int32 vw_stream_state_noise_levels(ctx, state, levels)
    static final MethodHandle VW_STREAM_STATE_NOISE_LEVELS = h("vw_stream_state_noise_levels",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS));
    static final MethodHandle VW_STREAM_STATE_SAMPLES = h("vw_stream_state_samples_processed", FunctionDescriptor.of(JAVA_LONG, ADDRESS));
    // int32 vw_streaming_denoise_batch(ctx, state, x, b, n, h, lh, g, lg, hr, lhr, gr, lgr, boundary, flags, method, soft,
    //                                  threshold_multiplier, noise_estimation, log_n, sqrt_2_ln_n, y, thresholds_out)
    static final MethodHandle VW_STREAMING_DENOISE = h("vw_streaming_denoise_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_DOUBLE, JAVA_INT, JAVA_DOUBLE, JAVA_DOUBLE,
            ADDRESS, ADDRESS));
    // int32 vw_batch_modwt_soa(ctx, soa_signals, batch, n, h, lh, g, lg, is_haar, soa_approx, soa_detail)
    static final MethodHandle VW_BATCH_MODWT_SOA = h("vw_batch_modwt_soa",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS,
            ADDRESS));
    // int32 vw_transpose(ctx, in, rows, cols, out)
    static final MethodHandle VW_TRANSPOSE = h("vw_transpose",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS));
    // int32 vw_swt_sparsify_batch(ctx, coeffs, b, n, levels, threshold, counts, offsets, indices, values)
    static final MethodHandle VW_SWT_SPARSIFY = h("vw_swt_sparsify_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, JAVA_DOUBLE, ADDRESS, ADDRESS, ADDRESS,
            ADDRESS));
    // int32 vw_swt_densify_batch(ctx, offsets, indices, values, b, n, levels, coeffs_out)
    static final MethodHandle VW_SWT_DENSIFY = h("vw_swt_densify_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS));
    // int32 vw_swt_universal_threshold_batch(ctx, coeffs, b, n, levels, sqrt_2_ln_n, thresholds)
    static final MethodHandle VW_UNIVERSAL_THRESHOLD = h("vw_swt_universal_threshold_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, JAVA_DOUBLE, ADDRESS));
    // int32 vw_threshold(ctx, in, count, threshold, soft, out)
    static final MethodHandle VW_THRESHOLD = h("vw_threshold",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_DOUBLE, JAVA_INT, ADDRESS));
    // int32 vw_modwt_convolve_batch(ctx, x, b, n, filter, filter_len, level, boundary, out)
    static final MethodHandle VW_CONVOLVE = h("vw_modwt_convolve_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT,
            ADDRESS));
    // int32 vw_first_non_finite(ctx, x, count, first_bad)
    static final MethodHandle VW_FIRST_NON_FINITE = h("vw_first_non_finite",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS));

    // int32 vw_host_alloc(ctx, bytes, out) / vw_host_free(ctx, ptr): page-locked host buffers (DMA at the PCIe rate)
    static final MethodHandle VW_HOST_ALLOC = h("vw_host_alloc", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS));
    static final MethodHandle VW_HOST_FREE = h("vw_host_free", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS));

    /** A page-locked off-heap segment of {@code bytes} bytes; release it with {@link #hostFree}. */
    public static MemorySegment hostAlloc(MemorySegment ctx, long bytes) {
        try (Arena a = Arena.ofConfined()) {
            MemorySegment out = a.allocate(ADDRESS);
            int st = (int) VW_HOST_ALLOC.invokeExact(ctx, bytes, out);
            if (st != VW_OK) {
                throw new OutOfMemoryError(lastError(ctx));
            }
            return out.get(ADDRESS, 0).reinterpret(bytes);
        } catch (RuntimeException | Error e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public static void hostFree(MemorySegment ctx, MemorySegment segment) {
        try {
            int st = (int) VW_HOST_FREE.invokeExact(ctx, segment);
            if (st != VW_OK) {
                throw new IllegalStateException(lastError(ctx));
            }
        } catch (RuntimeException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    private VwGpuNative() {}

    /** Opens a context on a HIP device; the returned segment is the opaque vw_ctx*. */
    public static MemorySegment init(int device) {
        try (Arena a = Arena.ofConfined()) {
            MemorySegment out = a.allocate(ADDRESS);
            int st = (int) VW_INIT.invokeExact(device, out);
            if (st != VW_OK) {
                throw new IllegalStateException("vw_init failed with status " + st + ": no usable HIP device");
            }
            return out.get(ADDRESS, 0);
        } catch (RuntimeException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public static String lastError(MemorySegment ctx) {
        try {
            MemorySegment p = (MemorySegment) VW_LAST_ERROR.invokeExact(ctx);
            return p.reinterpret(512).getString(0);
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }
}
