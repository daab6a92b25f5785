"""Shared helpers for the parity tests."""
import ctypes as C
import math

import numpy as np

from vectorwave_amd import _native as N
from vectorwave_amd.wavelets import get_wavelet

P, Z, S = 0, 1, 2


def signals(b, n, seed=42):
    """sin(2*pi*t/32) + 0.5*sin(2*pi*t/64) + 0.1*N(0,1): the shape of the reference's
    benchmark generator (src/test/.../swt/SWTBenchmarkTest.java:230-238)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    base = np.sin(2 * np.pi * t / 32.0) + 0.5 * np.sin(2 * np.pi * t / 64.0)
    return np.ascontiguousarray(base[None, :] + 0.1 * rng.standard_normal((b, n)))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _vp(a):
    return C.c_void_p(a.ctypes.data)


def _t(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def gpu_forward(ctx, x, wname, levels, boundary, flags=0):
    w = get_wavelet(wname)
    h, g = _t(w.dec_lo), _t(w.dec_hi)
    b, n = x.shape
    out = np.full((b, levels + 1, n), np.nan)
    bad = C.c_int64(-7)
    st = ctx.lib.vw_modwt_forward_batch(ctx.handle, _vp(x), b, n, levels, _dp(h), h.size, _dp(g), g.size, boundary,
                                        flags, _vp(out), C.byref(bad))
    return st, out, bad.value


def gpu_inverse(ctx, coeffs, wname, boundary, flags=0):
    w = get_wavelet(wname)
    hr, gr = _t(w.rec_lo), _t(w.rec_hi)
    b, jp1, n = coeffs.shape
    out = np.full((b, n), np.nan)
    bad = C.c_int64(-7)
    st = ctx.lib.vw_modwt_inverse_batch(ctx.handle, _vp(coeffs), b, n, jp1 - 1, _dp(hr), hr.size, _dp(gr), gr.size,
                                        boundary, flags, _vp(out), C.byref(bad))
    return st, out, bad.value


def gpu_denoise(ctx, x, wname, levels, boundary, threshold, soft):
    w = get_wavelet(wname)
    h, g, hr, gr = _t(w.dec_lo), _t(w.dec_hi), _t(w.rec_lo), _t(w.rec_hi)
    b, n = x.shape
    out = np.full((b, n), np.nan)
    thr = np.full(b, np.nan)
    st = ctx.lib.vw_swt_denoise_batch(ctx.handle, _vp(x), b, n, levels, _dp(h), h.size, _dp(g), g.size, _dp(hr),
                                      hr.size, _dp(gr), gr.size, boundary, float(threshold),
                                      math.sqrt(2 * math.log(n)), 1 if soft else 0, _vp(out), _vp(thr))
    return st, out, thr


def assert_bits_equal(a, b, what=""):
    """Bit-for-bit equality of two float64 arrays (NaN payloads and zero signs included)."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    ai, bi = a.view(np.uint64), b.view(np.uint64)
    if not np.array_equal(ai, bi):
        idx = np.argwhere(ai != bi)
        first = tuple(idx[0])
        raise AssertionError("%s: %d of %d elements differ; first at %s: got %r (%s) want %r (%s)" % (
            what, idx.shape[0], a.size, first, a[first], float(a[first]).hex(), b[first], float(b[first]).hex()))


# ---- device-resident helpers (torch tensors on the GPU, *_dev entry points) -------------------

def dev_forward(ctx, x_t, wname, levels, boundary, flags=0, out=None):
    import torch
    w = get_wavelet(wname)
    h, g = _t(w.dec_lo), _t(w.dec_hi)
    b, n = x_t.shape
    if out is None:
        out = torch.empty((b, levels + 1, n), dtype=torch.float64, device=x_t.device)
    sp = C.c_void_p(torch.cuda.current_stream(x_t.device).cuda_stream)
    st = ctx.lib.vw_modwt_forward_batch_dev(ctx.handle, C.c_void_p(x_t.data_ptr()), b, n, levels, _dp(h), h.size,
                                            _dp(g), g.size, boundary, flags, C.c_void_p(out.data_ptr()), None, sp)
    assert st == 0, ctx.last_error()
    return out


def dev_inverse(ctx, c_t, wname, boundary, flags=0, out=None):
    import torch
    w = get_wavelet(wname)
    hr, gr = _t(w.rec_lo), _t(w.rec_hi)
    b, jp1, n = c_t.shape
    if out is None:
        out = torch.empty((b, n), dtype=torch.float64, device=c_t.device)
    sp = C.c_void_p(torch.cuda.current_stream(c_t.device).cuda_stream)
    st = ctx.lib.vw_modwt_inverse_batch_dev(ctx.handle, C.c_void_p(c_t.data_ptr()), b, n, jp1 - 1, _dp(hr), hr.size,
                                            _dp(gr), gr.size, boundary, flags, C.c_void_p(out.data_ptr()), None, sp)
    assert st == 0, ctx.last_error()
    return out


def bind_dev_forward(ctx, x_t, wname, levels, boundary, flags, out):
    """dev_forward with the argument marshalling done once: returns a zero-argument callable (for timing loops in which
    the per-call cost of building ctypes arguments would otherwise dominate a launch-bound configuration)."""
    import torch
    w = get_wavelet(wname)
    h, g = _t(w.dec_lo), _t(w.dec_hi)
    b, n = x_t.shape
    sp = C.c_void_p(torch.cuda.current_stream(x_t.device).cuda_stream)
    fn = ctx.lib.vw_modwt_forward_batch_dev
    args = (ctx.handle, C.c_void_p(x_t.data_ptr()), C.c_int64(b), C.c_int64(n), C.c_int32(levels), _dp(h), C.c_int32(h.size), _dp(g),
            C.c_int32(g.size), C.c_int32(boundary), C.c_uint32(flags), C.c_void_p(out.data_ptr()), None, sp)
    keep = (h, g)

    def call():
        st = fn(*args)
        assert st == 0 and keep is not None, ctx.last_error()
    return call


def bind_dev_inverse(ctx, c_t, wname, boundary, flags, out):
    """dev_inverse with the argument marshalling done once (see bind_dev_forward)."""
    import torch
    w = get_wavelet(wname)
    hr, gr = _t(w.rec_lo), _t(w.rec_hi)
    b, jp1, n = c_t.shape
    sp = C.c_void_p(torch.cuda.current_stream(c_t.device).cuda_stream)
    fn = ctx.lib.vw_modwt_inverse_batch_dev
    args = (ctx.handle, C.c_void_p(c_t.data_ptr()), C.c_int64(b), C.c_int64(n), C.c_int32(jp1 - 1), _dp(hr), C.c_int32(hr.size), _dp(gr),
            C.c_int32(gr.size), C.c_int32(boundary), C.c_uint32(flags), C.c_void_p(out.data_ptr()), None, sp)
    keep = (hr, gr)

    def call():
        st = fn(*args)
        assert st == 0 and keep is not None, ctx.last_error()
    return call


def dev_denoise(ctx, x_t, wname, levels, boundary, threshold, soft, out=None):
    import torch
    w = get_wavelet(wname)
    h, g, hr, gr = _t(w.dec_lo), _t(w.dec_hi), _t(w.rec_lo), _t(w.rec_hi)
    b, n = x_t.shape
    if out is None:
        out = torch.empty((b, n), dtype=torch.float64, device=x_t.device)
    thr = torch.empty((b,), dtype=torch.float64, device=x_t.device)
    sp = C.c_void_p(torch.cuda.current_stream(x_t.device).cuda_stream)
    st = ctx.lib.vw_swt_denoise_batch_dev(ctx.handle, C.c_void_p(x_t.data_ptr()), b, n, levels, _dp(h), h.size, _dp(g),
                                          g.size, _dp(hr), hr.size, _dp(gr), gr.size, boundary, float(threshold),
                                          math.sqrt(2 * math.log(n)), 1 if soft else 0, C.c_void_p(out.data_ptr()),
                                          C.c_void_p(thr.data_ptr()), None, 0, sp)
    assert st == 0, ctx.last_error()
    return out, thr


def dev_wavelet_denoise(ctx, x_t, wname, levels, boundary, method, soft, out=None, fixed_threshold=0.0):
    """WaveletDenoiser on device-resident rows; levels == 0 selects the single-level form."""
    import torch
    w = get_wavelet(wname)
    h, g, hr, gr = _t(w.dec_lo), _t(w.dec_hi), _t(w.rec_lo), _t(w.rec_hi)
    b, n = x_t.shape
    nl = max(levels, 1)
    if out is None:
        out = torch.empty((b, n), dtype=torch.float64, device=x_t.device)
    thr = torch.empty((b, nl), dtype=torch.float64, device=x_t.device)
    sp = C.c_void_p(torch.cuda.current_stream(x_t.device).cuda_stream)
    flags = N.VW_FLAG_SINGLE_LEVEL if levels == 0 else 0
    st = ctx.lib.vw_wavelet_denoise_batch_dev(ctx.handle, C.c_void_p(x_t.data_ptr()), b, n, nl, _dp(h), h.size, _dp(g),
                                              g.size, _dp(hr), hr.size, _dp(gr), gr.size, boundary, flags, method,
                                              1 if soft else 0, float(fixed_threshold), math.log(n),
                                              C.c_void_p(out.data_ptr()), C.c_void_p(thr.data_ptr()), None, 0, sp)
    assert st == 0, ctx.last_error()
    return out, thr


def dev_signals(b, n, seed=42, device="cuda:0", noise=0.1):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.arange(n, dtype=torch.float64, device=device)
    base = torch.sin(2 * math.pi * t / 32.0) + 0.5 * torch.sin(2 * math.pi * t / 64.0)
    x = torch.empty((b, n), dtype=torch.float64, device=device)
    chunk = max(1, (256 << 20) // (n * 8))
    for r0 in range(0, b, chunk):
        r1 = min(b, r0 + chunk)
        x[r0:r1] = base[None] + noise * torch.randn((r1 - r0, n), dtype=torch.float64, device=device, generator=g)
    return x
