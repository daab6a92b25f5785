"""ctypes front-end of oracle/vw_oracle.c (the C restatement of the reference).

TEST INFRASTRUCTURE ONLY.  See vw_oracle.c for the reference citations of each
function.  `build()` compiles the shared object with gcc; `lib()` loads it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvw_oracle.so")

PERIODIC, ZERO_PADDING, SYMMETRIC = 0, 1, 2

_dp = C.POINTER(C.c_double)
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vw_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.vwo_estimate_noise_sigma.restype = C.c_double
        _lib.vwo_universal_threshold.restype = C.c_double
        _lib.vwo_first_non_finite.restype = C.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _conv(fn, signal, filt):
    s, f = _f64(signal), _f64(filt)
    out = np.empty_like(s)
    getattr(lib(), fn)(_p(s), C.c_int64(s.size), _p(f), C.c_int64(f.size), _p(out))
    return out


def circular_convolve(signal, filt):
    return _conv("vwo_circular_convolve", signal, filt)


def circular_convolve_scalar(signal, filt):
    return _conv("vwo_circular_convolve_scalar", signal, filt)


def circular_convolve_vector_order(signal, filt):
    return _conv("vwo_circular_convolve_vector_order", signal, filt)


def zero_padding_convolve(signal, filt):
    return _conv("vwo_zero_padding_convolve", signal, filt)


def symmetric_convolve(signal, filt):
    return _conv("vwo_symmetric_convolve", signal, filt)


def circular_convolve_level(signal, filt, level):
    s, f = _f64(signal), _f64(filt)
    out = np.empty_like(s)
    st = lib().vwo_circular_convolve_level(_p(s), C.c_int64(s.size), _p(f), C.c_int64(f.size), _p(out), C.c_int(level))
    if st != 0:
        raise ValueError("VAL_TOO_LARGE")
    return out


def max_levels(n, filter_len):
    return int(lib().vwo_max_levels(C.c_int64(n), C.c_int(filter_len)))


def modwt_forward_batch(x, wavelet, boundary=PERIODIC):
    """Single-level forward; returns coeffs [B][2][N] = (d_1, a_1)."""
    x = _f64(x)
    b, n = x.shape
    h, g = _f64(wavelet.dec_lo), _f64(wavelet.dec_hi)
    out = np.empty((b, 2, n))
    lib().vwo_modwt_forward_batch(_p(x), C.c_int64(b), C.c_int64(n), _p(h), C.c_int(h.size), _p(g),
                                  C.c_int(g.size), C.c_int(boundary), _p(out))
    return out


def modwt_inverse_batch(coeffs, wavelet, boundary=PERIODIC, batch_variant=1):
    c = _f64(coeffs)
    b, two, n = c.shape
    assert two == 2
    hr, gr = _f64(wavelet.rec_lo), _f64(wavelet.rec_hi)
    out = np.empty((b, n))
    lib().vwo_modwt_inverse_batch(_p(c), C.c_int64(b), C.c_int64(n), _p(hr), C.c_int(hr.size), _p(gr),
                                  C.c_int(gr.size), C.c_int(boundary), C.c_int(batch_variant), _p(out))
    return out


def decompose_batch(x, wavelet, levels, boundary=PERIODIC):
    """Multi-level forward; returns (status, coeffs [B][J+1][N] = d_1..d_J, a_J)."""
    x = _f64(x)
    b, n = x.shape
    h, g = _f64(wavelet.dec_lo), _f64(wavelet.dec_hi)
    out = np.empty((b, levels + 1, n))
    st = lib().vwo_multilevel_decompose_batch(_p(x), C.c_int64(b), C.c_int64(n), C.c_int(levels), _p(h),
                                             C.c_int(h.size), _p(g), C.c_int(g.size), C.c_int(boundary),
                                             _p(out))
    return int(st), out


def reconstruct_batch(coeffs, wavelet, boundary=PERIODIC):
    c = _f64(coeffs)
    b, jp1, n = c.shape
    hr, gr = _f64(wavelet.rec_lo), _f64(wavelet.rec_hi)
    out = np.empty((b, n))
    lib().vwo_multilevel_reconstruct_batch(_p(c), C.c_int64(b), C.c_int64(n), C.c_int(jp1 - 1), _p(hr),
                                           C.c_int(hr.size), _p(gr), C.c_int(gr.size), C.c_int(boundary),
                                           _p(out))
    return out


def swt_denoise_batch(x, wavelet, levels, threshold=-1.0, soft=True, boundary=PERIODIC):
    x = _f64(x)
    b, n = x.shape
    h, g = _f64(wavelet.dec_lo), _f64(wavelet.dec_hi)
    hr, gr = _f64(wavelet.rec_lo), _f64(wavelet.rec_hi)
    out = np.empty((b, n))
    st = lib().vwo_swt_denoise_batch(_p(x), C.c_int64(b), C.c_int64(n), C.c_int(levels), _p(h), C.c_int(h.size),
                                    _p(g), C.c_int(g.size), _p(hr), C.c_int(hr.size), _p(gr), C.c_int(gr.size),
                                    C.c_int(boundary), C.c_double(threshold), C.c_int(1 if soft else 0), _p(out))
    return int(st), out


def convert_to_soa(signals):
    """BatchSIMDMODWT.convertToSoA: [B][N] -> flat [N*B] with soa[t*B + b]."""
    x = _f64(signals)
    b, n = x.shape
    out = np.empty(b * n)
    lib().vwo_convert_to_soa(_p(x), C.c_int64(b), C.c_int64(n), _p(out))
    return out


def convert_from_soa(soa, batch, n):
    s = _f64(soa)
    out = np.empty((batch, n))
    lib().vwo_convert_from_soa(_p(s), C.c_int64(batch), C.c_int64(n), _p(out))
    return out


def batch_modwt_soa(soa, wavelet, batch, n):
    """BatchSIMDMODWT.batchMODWTSoA: returns (soaApprox, soaDetail)."""
    s = _f64(soa)
    h, g = _f64(wavelet.dec_lo), _f64(wavelet.dec_hi)
    a, d = np.empty(batch * n), np.empty(batch * n)
    lib().vwo_batch_modwt_soa(_p(s), _p(a), _p(d), _p(h), _p(g), C.c_int(h.size), C.c_int(1 if wavelet.name == "haar" else 0),
                              C.c_int64(batch), C.c_int64(n))
    return a, d


def soa_multilevel_roundtrip(soa, wavelet, batch, n, levels):
    """Multi-level forward + inverse in the reference's SIMD batch layout (soa[t*B + b], PERIODIC): the scheme
    of BatchSIMDMODWT's general path applied per level.  Returns (coeffs [(J+1)][N*B] planes d_1..d_J, a_J;
    reconstruction [N*B])."""
    s = _f64(soa)
    h, g = _f64(wavelet.dec_lo), _f64(wavelet.dec_hi)
    hr, gr = _f64(wavelet.rec_lo), _f64(wavelet.rec_hi)
    assert gr.size >= hr.size
    coeffs = np.empty((levels + 1, n * batch))
    scratch = np.empty(n * batch)
    out = np.empty(n * batch)
    lib().vwo_soa_multilevel_decompose(_p(s), C.c_int64(batch), C.c_int64(n), C.c_int(levels), _p(h), C.c_int(h.size),
                                       _p(g), C.c_int(g.size), _p(coeffs), _p(scratch))
    lib().vwo_soa_multilevel_reconstruct(_p(coeffs), C.c_int64(batch), C.c_int64(n), C.c_int(levels), _p(hr),
                                         C.c_int(hr.size), _p(gr), _p(out), _p(scratch))
    return coeffs, out


def parallel_max_levels(n, filter_len):
    return int(lib().vwo_parallel_max_levels(C.c_int64(n), C.c_int(filter_len)))


def parallel_decompose(x, wavelet, levels, boundary=PERIODIC):
    """ParallelMultiLevelMODWT.decompose for one signal: (status, coeffs [J+1][N])."""
    x = _f64(x)
    n = x.size
    h, g = _f64(wavelet.dec_lo), _f64(wavelet.dec_hi)
    out = np.empty((max(levels, 0) + 1, n))
    st = lib().vwo_parallel_decompose(_p(x), C.c_int64(n), C.c_int(levels), _p(h), C.c_int(h.size), _p(g), C.c_int(g.size),
                                      C.c_int(boundary), _p(out))
    return int(st), out


UNIVERSAL, SURE, MINIMAX, BAYES, FIXED = 0, 1, 2, 3, 4


def denoiser_threshold(coeffs, sigma, method):
    """WaveletDenoiser.calculateThreshold (denoising/WaveletDenoiser.java:385-427)."""
    c = _f64(coeffs)
    f = lib().vwo_denoiser_threshold
    f.restype = C.c_double
    return float(f(_p(c), C.c_int64(c.size), C.c_double(sigma), C.c_int(method)))


def wavelet_denoise_batch(x, wavelet, levels, method, soft=True, boundary=PERIODIC, fixed_threshold=0.0):
    """WaveletDenoiser.denoise / denoiseFixed (levels == 0) or denoiseMultiLevel (levels >= 1) per row.
    Returns (status, denoised [B][N], thresholds [B][max(levels, 1)])."""
    x = _f64(x)
    b, n = x.shape
    h, g = _f64(wavelet.dec_lo), _f64(wavelet.dec_hi)
    hr, gr = _f64(wavelet.rec_lo), _f64(wavelet.rec_hi)
    out = np.empty((b, n))
    thr = np.empty((b, max(levels, 1)))
    st = lib().vwo_wavelet_denoise_batch(_p(x), C.c_int64(b), C.c_int64(n), C.c_int(levels), _p(h), C.c_int(h.size),
                                        _p(g), C.c_int(g.size), _p(hr), C.c_int(hr.size), _p(gr), C.c_int(gr.size),
                                        C.c_int(boundary), C.c_int(method), C.c_double(fixed_threshold),
                                        C.c_int(1 if soft else 0), _p(out), _p(thr))
    return int(st), out, thr


NOISE_MAD, NOISE_STD, NOISE_FIXED = 0, 1, 2


class StreamingDenoiserState:
    """State of B parallel MODWTStreamingDenoiser instances (modwt/streaming/MODWTStreamingDenoiser.java:50-54):
    the |detail| ring per stream, the common ring position and the estimated noise level per stream."""

    def __init__(self, streams, window_size):
        self.window = np.zeros((streams, window_size))
        self.index = 0
        self.level = np.zeros(streams)
        self.samples_processed = 0


def streaming_denoise_batch(state, x, wavelet, method=UNIVERSAL, soft=True, multiplier=1.0, estimation=NOISE_MAD,
                            boundary=PERIODIC):
    """MODWTStreamingDenoiser.denoise (:93-127) of one block per stream; updates `state` in place.
    Returns (status, denoised [B][n], thresholds [B])."""
    x = _f64(x)
    b, n = x.shape
    h, g = _f64(wavelet.dec_lo), _f64(wavelet.dec_hi)
    hr, gr = _f64(wavelet.rec_lo), _f64(wavelet.rec_hi)
    out = np.empty((b, n))
    thr = np.full(b, np.nan)
    idx = C.c_int64(state.index)
    w = state.window.shape[1]
    st = lib().vwo_streaming_denoise_batch(_p(x), C.c_int64(b), C.c_int64(n), _p(h), C.c_int(h.size), _p(g), C.c_int(g.size),
                                          _p(hr), C.c_int(hr.size), _p(gr), C.c_int(gr.size), C.c_int(boundary),
                                          C.c_int(method), C.c_int(1 if soft else 0), C.c_double(multiplier),
                                          C.c_int(estimation), _p(state.window), C.c_int64(w), C.byref(idx), _p(state.level),
                                          _p(out), _p(thr))
    state.index = idx.value
    if st == 0:
        state.samples_processed += n
    return int(st), out, thr


class StreamingTransformState:
    """One MODWTStreamingTransformImpl (modwt/streaming/MODWTStreamingTransformImpl.java): sliding windows of
    bufferSize samples that overlap by filter length - 1, each transformed at a single level."""

    def __init__(self, wavelet, buffer_size, boundary=PERIODIC):
        self.wavelet, self.buffer_size, self.boundary = wavelet, int(buffer_size), boundary
        self.h, self.g = _f64(wavelet.dec_lo), _f64(wavelet.dec_hi)
        f = lib().vwo_stream_xform_create
        f.restype = C.c_void_p
        self.handle = C.c_void_p(f(C.c_int64(self.buffer_size), C.c_int(self.h.size)))

    def process(self, data):
        """Returns the blocks emitted while `data` was fed sample by sample: [blocks][2][bufferSize] = (detail, approx)."""
        d = _f64(data)
        hop = self.buffer_size - (self.h.size - 1)
        cap = d.size // hop + 2
        out = np.empty((cap, 2, self.buffer_size))
        f = lib().vwo_stream_xform_process
        f.restype = C.c_int64
        nb = f(self.handle, _p(d), C.c_int64(d.size), _p(self.h), C.c_int(self.h.size), _p(self.g), C.c_int(self.g.size),
               C.c_int(self.boundary), _p(out), C.c_int64(cap))
        assert nb <= cap
        return out[:nb].copy()

    def process_checked(self, data):
        """process() with the reference's failure rule: returns (blocks emitted before the failure, index of the sample
        whose window held a non-finite value or -1)."""
        d = _f64(data)
        hop = self.buffer_size - (self.h.size - 1)
        cap = d.size // max(hop, 1) + 2
        out = np.empty((cap, 2, self.buffer_size))
        failed = C.c_int64(-1)
        f = lib().vwo_stream_xform_process_checked
        f.restype = C.c_int64
        nb = f(self.handle, _p(d), C.c_int64(d.size), _p(self.h), C.c_int(self.h.size), _p(self.g), C.c_int(self.g.size),
               C.c_int(self.boundary), _p(out), C.c_int64(cap), C.byref(failed))
        assert nb <= cap
        return out[:nb].copy(), int(failed.value)

    def flush(self):
        out = np.empty((1, 2, self.buffer_size))
        nb = lib().vwo_stream_xform_flush(self.handle, _p(self.h), C.c_int(self.h.size), _p(self.g), C.c_int(self.g.size),
                                          C.c_int(self.boundary), _p(out))
        return out[:nb].copy()

    def reset(self):
        lib().vwo_stream_xform_reset(self.handle)

    def level(self):
        f = lib().vwo_stream_xform_level
        f.restype = C.c_int64
        return int(f(self.handle))

    def __del__(self):
        try:
            lib().vwo_stream_xform_destroy(self.handle)
        except Exception:
            pass


def stream_multilevel(data, wavelet, buffer_size, levels, boundary=PERIODIC, flush=True):
    """MultiLevelMODWTStreamingTransform over a whole stream: blocks [k][levels+1][bufferSize] (d_1..d_J, a_J)."""
    d = _f64(data)
    h, g = _f64(wavelet.dec_lo), _f64(wavelet.dec_hi)
    cap = d.size // buffer_size + 1
    out = np.empty((cap, levels + 1, buffer_size))
    f = lib().vwo_stream_multilevel
    f.restype = C.c_int64
    nb = f(_p(d), C.c_int64(d.size), C.c_int64(buffer_size), C.c_int(levels), _p(h), C.c_int(h.size), _p(g), C.c_int(g.size),
           C.c_int(boundary), C.c_int(1 if flush else 0), _p(out), C.c_int64(cap))
    if nb < 0:
        return int(-nb), None
    return 0, out[:nb].copy()


def sparse_vector(dense, threshold):
    """SWTResult.SparseVector (swt/SWTResult.java:267-290): (indices int32, values) of |v| > threshold."""
    d = _f64(dense)
    f = lib().vwo_sparse_vector
    f.restype = C.c_int64
    idx = np.empty(d.size, dtype=np.int32)
    val = np.empty(d.size)
    cnt = f(_p(d), C.c_int64(d.size), C.c_double(threshold), idx.ctypes.data_as(C.POINTER(C.c_int32)), _p(val))
    return idx[:cnt].copy(), val[:cnt].copy()


def sparse_to_dense(indices, values, n):
    idx = np.ascontiguousarray(indices, dtype=np.int32)
    val = _f64(values)
    out = np.empty(n)
    lib().vwo_sparse_to_dense(idx.ctypes.data_as(C.POINTER(C.c_int32)), _p(val), C.c_int64(idx.size), C.c_int64(n), _p(out))
    return out


def stream_mad(values):
    v = _f64(values)
    f = lib().vwo_stream_mad
    f.restype = C.c_double
    return float(f(_p(v), C.c_int64(v.size)))


def stream_std(values):
    v = _f64(values)
    f = lib().vwo_stream_std
    f.restype = C.c_double
    return float(f(_p(v), C.c_int64(v.size)))


def apply_threshold(coeffs, threshold, soft):
    c = _f64(coeffs).copy()
    lib().vwo_apply_threshold(_p(c), C.c_int64(c.size), C.c_double(threshold), C.c_int(1 if soft else 0))
    return c


def soft_threshold(coeffs, threshold):
    c = _f64(coeffs)
    out = np.empty_like(c)
    lib().vwo_soft_threshold(_p(c), C.c_int64(c.size), C.c_double(threshold), _p(out))
    return out


def hard_threshold(coeffs, threshold):
    c = _f64(coeffs)
    out = np.empty_like(c)
    lib().vwo_hard_threshold(_p(c), C.c_int64(c.size), C.c_double(threshold), _p(out))
    return out


def estimate_noise_sigma(coeffs):
    c = _f64(coeffs)
    return float(lib().vwo_estimate_noise_sigma(_p(c), C.c_int64(c.size)))


def universal_threshold(finest_detail):
    c = _f64(finest_detail)
    return float(lib().vwo_universal_threshold(_p(c), C.c_int64(c.size)))


def first_non_finite(x):
    c = _f64(x)
    return int(lib().vwo_first_non_finite(_p(c), C.c_int64(c.size)))
