"""The threading contract of the C ABI and the graph-capture use of the allocation-free entry point.

Reference contract (docs/ARCHITECTURE.md:130, modwt/MultiLevelMODWTTransform.java:127): the transform objects may be
used from several threads at once; mutable results are not thread-safe (modwt/MutableMultiLevelMODWTResult.java:14-15).
Here: one vw_ctx used from four host threads at once (the library serialises them on the context's mutex; ctypes
releases the GIL during a call), and two contexts on one GPU used at the same time -- every result bit-equal to the
CPU oracle.  Then vw_swt_denoise_batch_noalloc_dev recorded into a HIP graph (torch.cuda.graph) and replayed on changed
input, bit-equal to the eager call: what "callers that cannot allocate" comes down to."""
import ctypes as C
import math
import threading

import numpy as np
import pytest

from oracle import oracle as o
from util import _dp, _t, assert_bits_equal, gpu_denoise, gpu_forward, gpu_inverse, signals
from vectorwave_amd import _native as N
from vectorwave_amd.wavelets import get_wavelet

pytestmark = pytest.mark.gpu


def _jobs():
    """(name, run(ctx) -> got, want) pairs on different shapes, wavelets and boundary modes."""
    jobs = []

    def fwd(ctx, wn="db4", b=64, n=4096, j=6, mode=0, seed=11):
        x = signals(b, n, seed=seed)
        st, want = o.decompose_batch(x, get_wavelet(wn), j, mode)
        assert st == 0
        st, got, _ = gpu_forward(ctx, x, wn, j, mode)
        assert st == 0, ctx.last_error()
        return got, want

    def inv(ctx, wn="sym8", b=9, n=5000, j=5, mode=1, seed=12):
        x = signals(b, n, seed=seed)
        w = get_wavelet(wn)
        st, c = o.decompose_batch(x, w, j, mode)
        assert st == 0
        st, got, _ = gpu_inverse(ctx, c, wn, mode)
        assert st == 0, ctx.last_error()
        return got, o.reconstruct_batch(c, w, mode)

    def den(ctx, wn="db4", b=300, n=480, j=4, mode=0, seed=13):
        x = signals(b, n, seed=seed)
        st, want = o.swt_denoise_batch(x, get_wavelet(wn), j, -1.0, True, mode)
        assert st == 0
        st, got, _ = gpu_denoise(ctx, x, wn, j, mode, -1.0, True)
        assert st == 0, ctx.last_error()
        return got, want

    def bands(ctx, wn="haar", b=5, n=3000, j=4, seed=14):
        x = signals(b, n, seed=seed)
        w = get_wavelet(wn)
        st, c = o.decompose_batch(x, w, j, 0)
        assert st == 0
        mask = (1 << j) | 0b0101
        z = c.copy()
        for k in range(j):
            if not (mask >> k) & 1:
                z[:, k, :] = 0.0
        hr, gr = _t(w.rec_lo), _t(w.rec_hi)
        out = np.full((b, n), np.nan)
        st = ctx.lib.vw_modwt_inverse_bands_batch(ctx.handle, C.c_void_p(c.ctypes.data), b, n, j, _dp(hr), hr.size, _dp(gr), gr.size,
                                                  0, mask, C.c_void_p(out.ctypes.data))
        assert st == 0, ctx.last_error()
        return out, o.reconstruct_batch(z, w, 0)

    jobs.append(("forward", fwd))
    jobs.append(("inverse", inv))
    jobs.append(("denoise", den))
    jobs.append(("band mask", bands))
    return jobs


def _run_threads(targets):
    errors, results = [], {}

    def wrap(name, fn):
        try:
            for rep in range(3):
                results[(name, rep)] = fn()
        except BaseException as e:  # noqa: BLE001 -- reported below, in the main thread
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=wrap, args=(name, fn)) for name, fn in targets]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
        assert not t.is_alive(), "a worker thread is stuck inside the library"
    assert not errors, errors
    return results


def test_four_threads_on_one_context(ctx):
    res = _run_threads([(name, (lambda f=fn: f(ctx))) for name, fn in _jobs()])
    assert len(res) == 12
    for (name, rep), (got, want) in res.items():
        assert_bits_equal(got, want, "%s, repetition %d, four threads on one context" % (name, rep))


def test_two_contexts_on_one_gpu_at_the_same_time(ctx):
    other = N.Context(0)
    try:
        jobs = _jobs()
        targets = [("%s / ctx A" % name, (lambda f=fn: f(ctx))) for name, fn in jobs[:2]]
        targets += [("%s / ctx B" % name, (lambda f=fn: f(other))) for name, fn in jobs]
        res = _run_threads(targets)
        for (name, rep), (got, want) in res.items():
            assert_bits_equal(got, want, "%s, repetition %d, two contexts" % (name, rep))
    finally:
        other.close()


def _noalloc(ctx, x_t, out_t, thr_t, wname, levels, boundary, threshold, soft, stream):
    w = get_wavelet(wname)
    h, g, hr, gr = _t(w.dec_lo), _t(w.dec_hi), _t(w.rec_lo), _t(w.rec_hi)
    b, n = x_t.shape
    return ctx.lib.vw_swt_denoise_batch_noalloc_dev(ctx.handle, C.c_void_p(x_t.data_ptr()), b, n, levels, _dp(h), h.size, _dp(g), g.size,
                                                    _dp(hr), hr.size, _dp(gr), gr.size, boundary, float(threshold),
                                                    math.sqrt(2 * math.log(n)), 1 if soft else 0, C.c_void_p(out_t.data_ptr()),
                                                    C.c_void_p(thr_t.data_ptr()), C.c_void_p(stream.cuda_stream))


@pytest.mark.parametrize("wname,levels,b,n", [("db4", 6, 64, 16384),   # tile kernels, coefficient block in the static array
                                              ("db4", 4, 512, 480)])   # the streaming-block shape: one launch, no scratch
def test_noalloc_entry_point_inside_a_hip_graph(ctx, wname, levels, b, n):
    import torch
    dev = torch.device("cuda", 0)
    x_t = torch.from_numpy(signals(b, n, seed=21)).to(dev)
    out_t = torch.full((b, n), float("nan"), dtype=torch.float64, device=dev)
    thr_t = torch.full((b,), float("nan"), dtype=torch.float64, device=dev)
    # eager reference of the first input
    st = _noalloc(ctx, x_t, out_t, thr_t, wname, levels, 0, -1.0, True, torch.cuda.current_stream(dev))
    assert st == N.VW_OK, ctx.last_error()
    torch.cuda.synchronize()
    eager = out_t.cpu().numpy().copy()
    out_t.fill_(float("nan"))
    # record
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            st = _noalloc(ctx, x_t, out_t, thr_t, wname, levels, 0, -1.0, True, torch.cuda.current_stream(dev))
    assert st == N.VW_OK, ctx.last_error()
    torch.cuda.synchronize()
    assert bool(torch.isnan(out_t).all()), "recording must not run the kernels"
    # replay on the recorded input, then on changed input in the same buffers
    g.replay()
    torch.cuda.synchronize()
    assert_bits_equal(out_t.cpu().numpy(), eager, "first replay vs the eager call")
    for seed in (22, 23):
        x2 = signals(b, n, seed=seed)
        x_t.copy_(torch.from_numpy(x2).to(dev))
        out_t.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        st, want = o.swt_denoise_batch(x2, get_wavelet(wname), levels, -1.0, True, 0)
        assert st == 0
        assert_bits_equal(out_t.cpu().numpy(), want, "replay on changed input (seed %d) vs the oracle" % seed)
