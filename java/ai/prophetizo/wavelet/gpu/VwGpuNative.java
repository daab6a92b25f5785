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
 * <p>One handle per exported symbol, named like the symbol in upper case.  The handle block is generated from
 * the header (tools/gen_java_handles.py) and checked against it by tests/test_java_binding.py: arity and the
 * layout class of every parameter (ADDRESS / JAVA_INT / JAVA_LONG / JAVA_DOUBLE).
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
        VW_E_EMPTY = 6, VW_E_LENGTH_MISMATCH = 7, VW_E_BOUNDARY = 103, VW_E_LEVEL = 104, VW_E_UNSUPPORTED = 105,
        VW_E_NO_DEVICE = 900, VW_E_HIP = 901, VW_E_NOMEM = 902, VW_E_WORKSPACE = 903;
    // boundary modes (vw_boundary)
    public static final int BOUNDARY_PERIODIC = 0, BOUNDARY_ZERO_PADDING = 1, BOUNDARY_SYMMETRIC = 2;
    // flags
    public static final int FLAG_CHECK_FINITE = 1, FLAG_SYMMETRIC_BATCH_VARIANT = 2, FLAG_NO_LEVEL_CHECK = 4,
        FLAG_SINGLE_LEVEL = 8, FLAG_FMA = 16;
    // noise estimation of the streaming denoiser (vw_noise_estimation)
    public static final int NOISE_MAD = 0, NOISE_STD = 1, NOISE_FIXED = 2;

    private static MethodHandle h(String name, FunctionDescriptor fd) {
        return LINKER.downcallHandle(LIB.find(name).orElseThrow(() -> new UnsatisfiedLinkError(name)), fd);
    }

    // ---- BEGIN GENERATED HANDLES (tools/gen_java_handles.py; do not edit by hand) ----
    // int32_t vw_init(device, out_ctx)
    static final MethodHandle VW_INIT = h("vw_init", FunctionDescriptor.of(JAVA_INT, JAVA_INT, ADDRESS));
    // int32_t vw_destroy(ctx)
    static final MethodHandle VW_DESTROY = h("vw_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
    // const char * vw_last_error(ctx)
    static final MethodHandle VW_LAST_ERROR = h("vw_last_error", FunctionDescriptor.of(ADDRESS, ADDRESS));
    // const char * vw_performance_info(ctx)
    static final MethodHandle VW_PERFORMANCE_INFO = h("vw_performance_info", FunctionDescriptor.of(ADDRESS, ADDRESS));
    // int32_t vw_synchronize(ctx)
    static final MethodHandle VW_SYNCHRONIZE = h("vw_synchronize", FunctionDescriptor.of(JAVA_INT, ADDRESS));
    // int32_t vw_device_count()
    static final MethodHandle VW_DEVICE_COUNT = h("vw_device_count", FunctionDescriptor.of(JAVA_INT));
    // int32_t vw_set_plan_option(ctx, name, value)
    static final MethodHandle VW_SET_PLAN_OPTION = h("vw_set_plan_option",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG));
    // const char * vw_last_plan(ctx)
    static final MethodHandle VW_LAST_PLAN = h("vw_last_plan", FunctionDescriptor.of(ADDRESS, ADDRESS));
    // int32_t vw_host_alloc(ctx, bytes, out)
    static final MethodHandle VW_HOST_ALLOC = h("vw_host_alloc", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS));
    // int32_t vw_host_free(ctx, ptr)
    static final MethodHandle VW_HOST_FREE = h("vw_host_free", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS));
    // int32_t vw_max_levels(n, filter_len)
    static final MethodHandle VW_MAX_LEVELS = h("vw_max_levels", FunctionDescriptor.of(JAVA_INT, JAVA_LONG, JAVA_INT));
    // int32_t vw_modwt_forward_batch(ctx, x, b, n, levels, h, lh, g, lg, boundary, flags, coeffs, first_bad)
    static final MethodHandle VW_MODWT_FORWARD_BATCH = h("vw_modwt_forward_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS));
    // int32_t vw_modwt_forward_batch_dev(ctx, x_dev, b, n, levels, h, lh, g, lg, boundary, flags, coeffs_dev, first_bad_dev, stream)
    static final MethodHandle VW_MODWT_FORWARD_BATCH_DEV = h("vw_modwt_forward_batch_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS));
    // int32_t vw_modwt_inverse_batch(ctx, coeffs, b, n, levels, hr, lhr, gr, lgr, boundary, flags, x, first_bad)
    static final MethodHandle VW_MODWT_INVERSE_BATCH = h("vw_modwt_inverse_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS));
    // int32_t vw_modwt_inverse_batch_dev(ctx, coeffs_dev, b, n, levels, hr, lhr, gr, lgr, boundary, flags, x_dev, first_bad_dev, stream)
    static final MethodHandle VW_MODWT_INVERSE_BATCH_DEV = h("vw_modwt_inverse_batch_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS));
    // int32_t vw_modwt_inverse_bands_batch(ctx, coeffs, b, n, levels, hr, lhr, gr, lgr, boundary, band_mask, x)
    static final MethodHandle VW_MODWT_INVERSE_BANDS_BATCH = h("vw_modwt_inverse_bands_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS));
    // int32_t vw_modwt_inverse_bands_batch_dev(ctx, coeffs_dev, b, n, levels, hr, lhr, gr, lgr, boundary, band_mask, x_dev, stream)
    static final MethodHandle VW_MODWT_INVERSE_BANDS_BATCH_DEV = h("vw_modwt_inverse_bands_batch_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS));
    // int32_t vw_swt_denoise_batch(ctx, x, b, n, levels, h, lh, g, lg, hr, lhr, gr, lgr, boundary, threshold, sqrt_2_ln_n, soft, y, thresholds_out)
    static final MethodHandle VW_SWT_DENOISE_BATCH = h("vw_swt_denoise_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_DOUBLE, JAVA_DOUBLE, JAVA_INT, ADDRESS,
            ADDRESS));
    // int32_t vw_swt_denoise_batch_dev(ctx, x_dev, b, n, levels, h, lh, g, lg, hr, lhr, gr, lgr, boundary, threshold, sqrt_2_ln_n, soft, y_dev, thresholds_out_dev, workspace_dev, workspace_bytes, stream)
    static final MethodHandle VW_SWT_DENOISE_BATCH_DEV = h("vw_swt_denoise_batch_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_DOUBLE, JAVA_DOUBLE, JAVA_INT, ADDRESS, ADDRESS,
            ADDRESS, JAVA_LONG, ADDRESS));
    // size_t vw_swt_denoise_workspace_bytes(b, n, levels)
    static final MethodHandle VW_SWT_DENOISE_WORKSPACE_BYTES = h("vw_swt_denoise_workspace_bytes",
        FunctionDescriptor.of(JAVA_LONG, JAVA_LONG, JAVA_LONG, JAVA_INT));
    // int32_t vw_swt_denoise_batch_noalloc_dev(ctx, x_dev, b, n, levels, h, lh, g, lg, hr, lhr, gr, lgr, boundary, threshold, sqrt_2_ln_n, soft, y_dev, thresholds_out_dev, stream)
    static final MethodHandle VW_SWT_DENOISE_BATCH_NOALLOC_DEV = h("vw_swt_denoise_batch_noalloc_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_DOUBLE, JAVA_DOUBLE, JAVA_INT, ADDRESS, ADDRESS,
            ADDRESS));
    // int32_t vw_wavelet_denoise_batch(ctx, x, b, n, levels, h, lh, g, lg, hr, lhr, gr, lgr, boundary, flags, method, soft, fixed_threshold, log_n, y, thresholds_out)
    static final MethodHandle VW_WAVELET_DENOISE_BATCH = h("vw_wavelet_denoise_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_DOUBLE,
            JAVA_DOUBLE, ADDRESS, ADDRESS));
    // int32_t vw_wavelet_denoise_batch_dev(ctx, x_dev, b, n, levels, h, lh, g, lg, hr, lhr, gr, lgr, boundary, flags, method, soft, fixed_threshold, log_n, y_dev, thresholds_out_dev, workspace_dev, workspace_bytes, stream)
    static final MethodHandle VW_WAVELET_DENOISE_BATCH_DEV = h("vw_wavelet_denoise_batch_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_DOUBLE,
            JAVA_DOUBLE, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS));
    // size_t vw_wavelet_denoise_workspace_bytes(b, n, levels)
    static final MethodHandle VW_WAVELET_DENOISE_WORKSPACE_BYTES = h("vw_wavelet_denoise_workspace_bytes",
        FunctionDescriptor.of(JAVA_LONG, JAVA_LONG, JAVA_LONG, JAVA_INT));
    // int32_t vw_denoiser_thresholds_batch(ctx, coeffs, b, n, levels, method, level_scaled, log_n, thresholds)
    static final MethodHandle VW_DENOISER_THRESHOLDS_BATCH = h("vw_denoiser_thresholds_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_DOUBLE,
            ADDRESS));
    // int32_t vw_denoiser_thresholds_batch_dev(ctx, coeffs_dev, b, n, levels, method, level_scaled, log_n, thresholds_dev, sigma_scratch_dev, stream)
    static final MethodHandle VW_DENOISER_THRESHOLDS_BATCH_DEV = h("vw_denoiser_thresholds_batch_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_DOUBLE,
            ADDRESS, ADDRESS, ADDRESS));
    // int32_t vw_stream_state_create(ctx, streams, window_size, out)
    static final MethodHandle VW_STREAM_STATE_CREATE = h("vw_stream_state_create",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS));
    // int32_t vw_stream_state_destroy(ctx, state)
    static final MethodHandle VW_STREAM_STATE_DESTROY = h("vw_stream_state_destroy",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS));
    // int32_t vw_stream_state_noise_levels(ctx, state, levels)
    static final MethodHandle VW_STREAM_STATE_NOISE_LEVELS = h("vw_stream_state_noise_levels",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS));
    // int32_t vw_stream_state_window(ctx, state, window, window_index)
    static final MethodHandle VW_STREAM_STATE_WINDOW = h("vw_stream_state_window",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
    // int64_t vw_stream_state_samples_processed(state)
    static final MethodHandle VW_STREAM_STATE_SAMPLES_PROCESSED = h("vw_stream_state_samples_processed",
        FunctionDescriptor.of(JAVA_LONG, ADDRESS));
    // int32_t vw_streaming_denoise_batch(ctx, state, x, b, n, h, lh, g, lg, hr, lhr, gr, lgr, boundary, flags, method, soft, threshold_multiplier, noise_estimation, log_n, sqrt_2_ln_n, y, thresholds_out)
    static final MethodHandle VW_STREAMING_DENOISE_BATCH = h("vw_streaming_denoise_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_DOUBLE, JAVA_INT,
            JAVA_DOUBLE, JAVA_DOUBLE, ADDRESS, ADDRESS));
    // int32_t vw_streaming_denoise_batch_dev(ctx, state, x_dev, b, n, h, lh, g, lg, hr, lhr, gr, lgr, boundary, flags, method, soft, threshold_multiplier, noise_estimation, log_n, sqrt_2_ln_n, y_dev, thresholds_out_dev, workspace_dev, workspace_bytes, stream)
    static final MethodHandle VW_STREAMING_DENOISE_BATCH_DEV = h("vw_streaming_denoise_batch_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS, JAVA_INT, ADDRESS,
            JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_DOUBLE, JAVA_INT,
            JAVA_DOUBLE, JAVA_DOUBLE, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS));
    // int32_t vw_batch_modwt_soa(ctx, soa_signals, batch, n, h, lh, g, lg, is_haar, soa_approx, soa_detail)
    static final MethodHandle VW_BATCH_MODWT_SOA = h("vw_batch_modwt_soa",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT,
            JAVA_INT, ADDRESS, ADDRESS));
    // int32_t vw_batch_modwt_soa_dev(ctx, soa_signals_dev, batch, n, h, lh, g, lg, is_haar, soa_approx_dev, soa_detail_dev, stream)
    static final MethodHandle VW_BATCH_MODWT_SOA_DEV = h("vw_batch_modwt_soa_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT,
            JAVA_INT, ADDRESS, ADDRESS, ADDRESS));
    // int32_t vw_transpose(ctx, in, rows, cols, out)
    static final MethodHandle VW_TRANSPOSE = h("vw_transpose",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS));
    // int32_t vw_transpose_dev(ctx, in_dev, rows, cols, out_dev, stream)
    static final MethodHandle VW_TRANSPOSE_DEV = h("vw_transpose_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS, ADDRESS));
    // int32_t vw_swt_universal_threshold_batch(ctx, coeffs, b, n, levels, sqrt_2_ln_n, thresholds)
    static final MethodHandle VW_SWT_UNIVERSAL_THRESHOLD_BATCH = h("vw_swt_universal_threshold_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, JAVA_DOUBLE, ADDRESS));
    // int32_t vw_swt_universal_threshold_batch_dev(ctx, coeffs_dev, b, n, levels, sqrt_2_ln_n, thresholds_dev, stream)
    static final MethodHandle VW_SWT_UNIVERSAL_THRESHOLD_BATCH_DEV = h("vw_swt_universal_threshold_batch_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, JAVA_DOUBLE, ADDRESS, ADDRESS));
    // int32_t vw_swt_sparsify_batch(ctx, coeffs, b, n, levels, threshold, counts, offsets, indices, values)
    static final MethodHandle VW_SWT_SPARSIFY_BATCH = h("vw_swt_sparsify_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, JAVA_DOUBLE, ADDRESS, ADDRESS,
            ADDRESS, ADDRESS));
    // int32_t vw_swt_sparsify_batch_dev(ctx, coeffs_dev, b, n, levels, threshold, counts_dev, offsets_dev, indices_dev, values_dev, stream)
    static final MethodHandle VW_SWT_SPARSIFY_BATCH_DEV = h("vw_swt_sparsify_batch_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, JAVA_DOUBLE, ADDRESS, ADDRESS,
            ADDRESS, ADDRESS, ADDRESS));
    // int32_t vw_swt_densify_batch(ctx, offsets, indices, values, b, n, levels, coeffs_out)
    static final MethodHandle VW_SWT_DENSIFY_BATCH = h("vw_swt_densify_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS));
    // int32_t vw_swt_densify_batch_dev(ctx, offsets_dev, indices_dev, values_dev, b, n, levels, coeffs_out_dev, stream)
    static final MethodHandle VW_SWT_DENSIFY_BATCH_DEV = h("vw_swt_densify_batch_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS,
            ADDRESS));
    // int32_t vw_threshold(ctx, in, count, threshold, soft, out)
    static final MethodHandle VW_THRESHOLD = h("vw_threshold",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_DOUBLE, JAVA_INT, ADDRESS));
    // int32_t vw_threshold_dev(ctx, in_dev, count, threshold, soft, out_dev, stream)
    static final MethodHandle VW_THRESHOLD_DEV = h("vw_threshold_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_DOUBLE, JAVA_INT, ADDRESS, ADDRESS));
    // int32_t vw_modwt_convolve_batch(ctx, x, b, n, filter, filter_len, level, boundary, out)
    static final MethodHandle VW_MODWT_CONVOLVE_BATCH = h("vw_modwt_convolve_batch",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT,
            ADDRESS));
    // int32_t vw_modwt_convolve_batch_dev(ctx, x_dev, b, n, filter, filter_len, level, boundary, out_dev, stream)
    static final MethodHandle VW_MODWT_CONVOLVE_BATCH_DEV = h("vw_modwt_convolve_batch_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, JAVA_LONG, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT,
            ADDRESS, ADDRESS));
    // int32_t vw_first_non_finite(ctx, x, count, first_bad)
    static final MethodHandle VW_FIRST_NON_FINITE = h("vw_first_non_finite",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS));
    // int32_t vw_first_non_finite_dev(ctx, x_dev, count, first_bad_dev, stream)
    static final MethodHandle VW_FIRST_NON_FINITE_DEV = h("vw_first_non_finite_dev",
        FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS));
    // ---- END GENERATED HANDLES ----

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

    /** vw_last_error: the message of the last failure on the context. */
    public static String lastError(MemorySegment ctx) {
        try {
            MemorySegment p = (MemorySegment) VW_LAST_ERROR.invokeExact(ctx);
            return p.reinterpret(512).getString(0);
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /** vw_performance_info: description of the execution back end. */
    public static String performanceInfo(MemorySegment ctx) {
        try {
            MemorySegment p = (MemorySegment) VW_PERFORMANCE_INFO.invokeExact(ctx);
            return p.reinterpret(512).getString(0);
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /** A page-locked off-heap segment of {@code bytes} bytes (vw_host_alloc); release it with {@link #hostFree}. */
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
}
