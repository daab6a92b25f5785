"""ctypes binding of libvwgpu.so (the C ABI declared in include/vwgpu.h).

The library is the only compute path of this package: if it cannot be loaded,
or no HIP device is present when a context is requested, the failure is raised
to the caller.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VW_LIBVWGPU lets the tuning harness swap in an experimental build of the same ABI
LIB_PATH = os.environ.get("VW_LIBVWGPU") or os.path.join(_HERE, "libvwgpu.so")

# status codes (include/vwgpu.h)
VW_OK = 0
VW_E_NULL = 1
VW_E_NONFINITE = 3
VW_E_NOT_POSITIVE = 4
VW_E_TOO_LARGE = 5
VW_E_EMPTY = 6
VW_E_LENGTH_MISMATCH = 7
VW_E_BOUNDARY = 103
VW_E_LEVEL = 104
VW_E_UNSUPPORTED = 105
VW_E_NO_DEVICE = 900
VW_E_HIP = 901
VW_E_NOMEM = 902
VW_E_WORKSPACE = 903

VW_FLAG_CHECK_FINITE = 1
VW_FLAG_SYMMETRIC_BATCH_VARIANT = 2
VW_FLAG_NO_LEVEL_CHECK = 4
VW_FLAG_SINGLE_LEVEL = 8
VW_FLAG_FMA = 16

VW_THRESHOLD_UNIVERSAL = 0
VW_THRESHOLD_SURE = 1
VW_THRESHOLD_MINIMAX = 2
VW_THRESHOLD_BAYES = 3
VW_THRESHOLD_FIXED = 4

VW_NOISE_MAD = 0
VW_NOISE_STD = 1
VW_NOISE_FIXED = 2

_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol of include/vwgpu.h
SIGNATURES = {
    "vw_init": (C.c_int32, [C.c_int32, C.POINTER(_vp)]),
    "vw_destroy": (C.c_int32, [_vp]),
    "vw_last_error": (C.c_char_p, [_vp]),
    "vw_performance_info": (C.c_char_p, [_vp]),
    "vw_synchronize": (C.c_int32, [_vp]),
    "vw_set_plan_option": (C.c_int32, [_vp, C.c_char_p, C.c_int64]),
    "vw_last_plan": (C.c_char_p, [_vp]),
    "vw_device_count": (C.c_int32, []),
    "vw_max_levels": (C.c_int32, [C.c_int64, C.c_int32]),
    "vw_modwt_forward_batch": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, _dp, C.c_int32, _dp, C.c_int32,
                                           C.c_int32, C.c_uint32, _vp, _i64p]),
    "vw_modwt_forward_batch_dev": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, _dp, C.c_int32, _dp,
                                               C.c_int32, C.c_int32, C.c_uint32, _vp, _vp, _vp]),
    "vw_modwt_inverse_batch": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, _dp, C.c_int32, _dp, C.c_int32,
                                           C.c_int32, C.c_uint32, _vp, _i64p]),
    "vw_modwt_inverse_batch_dev": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, _dp, C.c_int32, _dp,
                                               C.c_int32, C.c_int32, C.c_uint32, _vp, _vp, _vp]),
    "vw_modwt_inverse_bands_batch": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, _dp, C.c_int32, _dp,
                                                 C.c_int32, C.c_int32, C.c_uint32, _vp]),
    "vw_modwt_inverse_bands_batch_dev": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, _dp, C.c_int32, _dp,
                                                     C.c_int32, C.c_int32, C.c_uint32, _vp, _vp]),
    "vw_swt_denoise_batch": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, _dp, C.c_int32, _dp, C.c_int32,
                                         _dp, C.c_int32, _dp, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32,
                                         _vp, _vp]),
    "vw_swt_denoise_batch_dev": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, _dp, C.c_int32, _dp, C.c_int32,
                                             _dp, C.c_int32, _dp, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                             C.c_int32, _vp, _vp, _vp, C.c_size_t, _vp]),
    "vw_swt_denoise_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32]),
    "vw_swt_denoise_batch_noalloc_dev": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, _dp, C.c_int32, _dp,
                                                     C.c_int32, _dp, C.c_int32, _dp, C.c_int32, C.c_int32, C.c_double,
                                                     C.c_double, C.c_int32, _vp, _vp, _vp]),
    "vw_wavelet_denoise_batch": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, _dp, C.c_int32, _dp, C.c_int32,
                                             _dp, C.c_int32, _dp, C.c_int32, C.c_int32, C.c_uint32, C.c_int32, C.c_int32,
                                             C.c_double, C.c_double, _vp, _vp]),
    "vw_wavelet_denoise_batch_dev": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, _dp, C.c_int32, _dp, C.c_int32,
                                                 _dp, C.c_int32, _dp, C.c_int32, C.c_int32, C.c_uint32, C.c_int32,
                                                 C.c_int32, C.c_double, C.c_double, _vp, _vp, _vp, C.c_size_t, _vp]),
    "vw_wavelet_denoise_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32]),
    "vw_denoiser_thresholds_batch": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                                 C.c_double, _vp]),
    "vw_denoiser_thresholds_batch_dev": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                                     C.c_double, _vp, _vp, _vp]),
    "vw_batch_modwt_soa": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, _dp, C.c_int32, _dp, C.c_int32, C.c_int32, _vp, _vp]),
    "vw_batch_modwt_soa_dev": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, _dp, C.c_int32, _dp, C.c_int32, C.c_int32, _vp, _vp,
                                           _vp]),
    "vw_stream_state_create": (C.c_int32, [_vp, C.c_int64, C.c_int64, C.POINTER(_vp)]),
    "vw_stream_state_destroy": (C.c_int32, [_vp, _vp]),
    "vw_stream_state_noise_levels": (C.c_int32, [_vp, _vp, _vp]),
    "vw_stream_state_window": (C.c_int32, [_vp, _vp, _vp, _i64p]),
    "vw_stream_state_samples_processed": (C.c_int64, [_vp]),
    "vw_streaming_denoise_batch": (C.c_int32, [_vp, _vp, _vp, C.c_int64, C.c_int64, _dp, C.c_int32, _dp, C.c_int32, _dp,
                                               C.c_int32, _dp, C.c_int32, C.c_int32, C.c_uint32, C.c_int32, C.c_int32,
                                               C.c_double, C.c_int32, C.c_double, C.c_double, _vp, _vp]),
    "vw_streaming_denoise_batch_dev": (C.c_int32, [_vp, _vp, _vp, C.c_int64, C.c_int64, _dp, C.c_int32, _dp, C.c_int32,
                                                   _dp, C.c_int32, _dp, C.c_int32, C.c_int32, C.c_uint32, C.c_int32,
                                                   C.c_int32, C.c_double, C.c_int32, C.c_double, C.c_double, _vp, _vp,
                                                   _vp, C.c_size_t, _vp]),
    "vw_swt_sparsify_batch": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, C.c_double, _vp, _vp, _vp, _vp]),
    "vw_swt_sparsify_batch_dev": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "vw_swt_densify_batch": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int32, _vp]),
    "vw_swt_densify_batch_dev": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int32, _vp, _vp]),
    "vw_host_alloc": (C.c_int32, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "vw_host_free": (C.c_int32, [_vp, _vp]),
    "vw_transpose": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, _vp]),
    "vw_transpose_dev": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, _vp, _vp]),
    "vw_swt_universal_threshold_batch": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, C.c_double, _vp]),
    "vw_swt_universal_threshold_batch_dev": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int32, C.c_double, _vp,
                                                         _vp]),
    "vw_threshold": (C.c_int32, [_vp, _vp, C.c_int64, C.c_double, C.c_int32, _vp]),
    "vw_threshold_dev": (C.c_int32, [_vp, _vp, C.c_int64, C.c_double, C.c_int32, _vp, _vp]),
    "vw_modwt_convolve_batch": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, _dp, C.c_int32, C.c_int32, C.c_int32,
                                            _vp]),
    "vw_modwt_convolve_batch_dev": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, _dp, C.c_int32, C.c_int32, C.c_int32,
                                                _vp, _vp]),
    "vw_first_non_finite": (C.c_int32, [_vp, _vp, C.c_int64, _i64p]),
    "vw_first_non_finite_dev": (C.c_int32, [_vp, _vp, C.c_int64, _vp, _vp]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    """libvwgpu.so is missing or no GPU is usable.  Never swallowed: the package has no CPU path."""


def _share_torch_hip_runtime():
    """One HIP/HSA runtime per process.  PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64
    and ask for them by file name; libvwgpu.so asks for the soname.  If libvwgpu.so is loaded first it
    binds the system runtime, torch then loads its bundled copy as a second runtime, and that one finds
    no device.  When a torch installation is present its copies are therefore loaded first (without
    importing torch), so both bind the same runtime whatever the import order.  Without torch (the Java
    host) nothing happens and the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load():
    """Load libvwgpu.so and declare every prototype.  Raises NativeLibraryError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            "%s not found: build it with `make -C vectorwave_amd/csrc` (hipcc, gfx950). "
            "vectorwave_amd has no CPU fallback." % LIB_PATH)
    _share_torch_hip_runtime()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing HIP runtime etc.
        raise NativeLibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class _ContextState:
    """What the finalizers of page-locked arrays share with their Context: the native handle stays alive until
    both the Context has been closed and the last such array has been released, whichever comes last, so that
    vw_host_free never sees a destroyed context and an array never outlives its memory."""

    def __init__(self, lib, handle):
        self.lib = lib
        self.handle = handle
        self.outstanding = 0
        self.closed = False

    def release_pinned(self, address):
        if self.handle is None:
            return
        self.lib.vw_host_free(self.handle, C.c_void_p(address))
        self.outstanding -= 1
        self._maybe_destroy()

    def close(self):
        self.closed = True
        self._maybe_destroy()

    def _maybe_destroy(self):
        if self.closed and self.outstanding == 0 and self.handle is not None:
            self.lib.vw_destroy(self.handle)
            self.handle = None


class Context:
    """One vw_ctx: a HIP device plus the library-owned staging buffers and stream."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = _vp()
        st = self.lib.vw_init(int(device), C.byref(h))
        if st != VW_OK:
            raise NativeLibraryError(
                "vw_init(device=%d) failed with status %d: no usable HIP device. "
                "vectorwave_amd has no CPU fallback." % (device, st))
        self.handle = h
        self.device = device
        self._state = _ContextState(self.lib, h)

    def last_error(self) -> str:
        return self.lib.vw_last_error(self.handle).decode()

    def performance_info(self) -> str:
        return self.lib.vw_performance_info(self.handle).decode()

    def last_plan(self):
        """vw_last_plan: the kernels the last forward and the last inverse call enqueued."""
        return self.lib.vw_last_plan(self.handle).decode()

    def set_plan_option(self, name: str, value: int) -> None:
        """vw_set_plan_option: choose among equivalent launch plans (tests and tuning); 'reset' restores defaults."""
        st = self.lib.vw_set_plan_option(self.handle, name.encode(), int(value))
        if st != VW_OK:
            raise ValueError(self.last_error())

    def plan_options(self, **options):
        """Context manager: apply plan options for the duration of a `with` block, then reset all of them."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            try:
                for k, v in options.items():
                    self.set_plan_option(k, v)
                yield self
            finally:
                self.set_plan_option("reset", 0)
        return scope()

    def host_array(self, shape, dtype="float64"):
        """A NumPy array over page-locked host memory (vw_host_alloc): the host-pointer entry points copy such
        buffers by DMA at the PCIe rate.  The memory is released when the array is garbage-collected; a Context
        closed before that keeps its native handle until the last such array is gone."""
        import numpy as np
        dt = np.dtype(dtype)
        count = int(np.prod(shape))
        if count == 0:
            return np.empty(shape, dtype=dt)
        ptr = C.c_void_p()
        st = self.lib.vw_host_alloc(self.handle, count * dt.itemsize, C.byref(ptr))
        if st != VW_OK:
            raise MemoryError(self.last_error())
        buf = (C.c_char * (count * dt.itemsize)).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=dt).reshape(shape)
        state, address = self._state, ptr.value
        state.outstanding += 1
        import weakref
        weakref.finalize(buf, state.release_pinned, address)
        return arr

    def synchronize(self):
        return self.lib.vw_synchronize(self.handle)

    def close(self):
        """Release the context.  Further calls through this object fail; the native handle itself is destroyed
        once no page-locked array obtained from host_array() is alive any more."""
        if self.handle is not None:
            self.handle = None
            self._state.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device: int = 0) -> Context:
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = Context(device)
        _default_ctx[device] = ctx
    return ctx
