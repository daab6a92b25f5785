"""The allocation-free denoise entry point against the allocating one.

vw_swt_denoise_batch_noalloc_dev keeps the coefficient block and the thresholds in a static per-GPU array instead of
a workspace; kernels and results are those of vw_swt_denoise_batch_dev.  Two calls in a row, the second on a
smaller input (it must not see anything the first one left in the array), every result bit-identical to the
allocating entry point and to the oracle; sizes beyond the static array are refused before anything is enqueued."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import oracle as o
from util import _dp, _t, assert_bits_equal, dev_denoise, signals
from vectorwave_amd import _native as N
from vectorwave_amd.wavelets import get_wavelet

pytestmark = pytest.mark.gpu


def noalloc_denoise(ctx, x_t, wname, levels, boundary, threshold, soft):
    import torch
    w = get_wavelet(wname)
    h, g, hr, gr = _t(w.dec_lo), _t(w.dec_hi), _t(w.rec_lo), _t(w.rec_hi)
    b, n = x_t.shape
    out = torch.full((b, n), float("nan"), dtype=torch.float64, device=x_t.device)
    thr = torch.full((b,), float("nan"), dtype=torch.float64, device=x_t.device)
    sp = C.c_void_p(torch.cuda.current_stream(x_t.device).cuda_stream)
    st = ctx.lib.vw_swt_denoise_batch_noalloc_dev(ctx.handle, C.c_void_p(x_t.data_ptr()), b, n, levels, _dp(h), h.size, _dp(g),
                                                  g.size, _dp(hr), hr.size, _dp(gr), gr.size, boundary, float(threshold),
                                                  math.sqrt(2 * math.log(n)), 1 if soft else 0, C.c_void_p(out.data_ptr()),
                                                  C.c_void_p(thr.data_ptr()), sp)
    return st, out, thr


@pytest.mark.parametrize("wname,levels,boundary,first,second,threshold,soft", [
    ("db4", 4, 0, (8192, 480), (1000, 480), -1.0, True),     # the streaming-block shape: row-resident kernel, no scratch
    ("db4", 4, 0, (1000, 4096), (100, 2500), -1.0, True),    # rows too long for it: 156 MiB of the array, then a small call
    ("db4", 6, 0, (64, 16384), (3, 5000), -1.0, True),       # tile kernels, then a smaller batch of shorter rows
    ("sym8", 8, 1, (6, 20000), (2, 9000), 0.25, False),      # long-context kernels, fixed threshold, hard rule
    ("coif3", 7, 0, (5, 12288), (5, 3000), -1.0, True),
    ("db4", 3, 2, (16, 700), (4, 64), -1.0, True),           # SYMMETRIC: generic per-level kernels, intermediate plane
])
def test_two_calls_in_a_row_match_the_allocating_entry_point(ctx, wname, levels, boundary, first, second, threshold, soft):
    import torch
    w = get_wavelet(wname)
    for k, (b, n) in enumerate((first, second)):
        x = signals(b, n, seed=100 * k + n % 97, )
        x_t = torch.from_numpy(x).cuda()
        st, got, thr = noalloc_denoise(ctx, x_t, wname, levels, boundary, threshold, soft)
        assert st == N.VW_OK, ctx.last_error()
        want, want_thr = dev_denoise(ctx, x_t, wname, levels, boundary, threshold, soft)
        torch.cuda.synchronize()
        assert_bits_equal(got.cpu().numpy(), want.cpu().numpy(), "call %d output vs vw_swt_denoise_batch_dev" % k)
        assert_bits_equal(thr.cpu().numpy(), want_thr.cpu().numpy(), "call %d thresholds" % k)
        if b * n <= 600000:
            st, ref = o.swt_denoise_batch(x, w, levels, threshold, soft, boundary)
            assert st == 0
            assert_bits_equal(got.cpu().numpy(), ref, "call %d output vs oracle" % k)


def test_multi_pass_plan_borrows_its_plane_from_the_array(ctx, plan):
    """With the long-context kernels switched off a db20 J=8 cascade (context 39 * 255 samples, beyond the largest
    halo shape) takes two passes; their intermediate plane then comes out of what the coefficient block leaves of
    the static array."""
    import torch
    plan("no_sdry", 1)
    x_t = torch.from_numpy(signals(4, 16384, seed=5)).cuda()
    st, got, thr = noalloc_denoise(ctx, x_t, "db20", 8, 0, -1.0, True)
    assert st == N.VW_OK, ctx.last_error()
    want, want_thr = dev_denoise(ctx, x_t, "db20", 8, 0, -1.0, True)
    assert_bits_equal(got.cpu().numpy(), want.cpu().numpy(), "multi-pass output")
    assert_bits_equal(thr.cpu().numpy(), want_thr.cpu().numpy(), "multi-pass thresholds")


def test_sizes_beyond_the_static_array_are_refused(ctx):
    import torch
    limit = 160 << 20
    b, n, levels = 1024, 4096, 4  # 1024 * 5 * 4096 * 8 = 160 MiB of coefficients + thresholds: just over
    assert ctx.lib.vw_swt_denoise_workspace_bytes(b, n, levels) > limit
    x_t = torch.zeros((b, n), dtype=torch.float64, device="cuda")
    st, out, _ = noalloc_denoise(ctx, x_t, "db4", levels, 0, -1.0, True)
    assert st == N.VW_E_WORKSPACE
    assert "VW_NOALLOC_SCRATCH_BYTES" in ctx.last_error()
    torch.cuda.synchronize()
    assert bool(torch.isnan(out).all()), "a refused call must not enqueue anything"
    # a multi-pass plan whose plane does not fit behind its coefficient block is refused too
    with ctx.plan_options(no_sdry=1):
        b, n, levels = 130, 16384, 8
        assert ctx.lib.vw_swt_denoise_workspace_bytes(b, n, levels) < limit
        x_t = torch.from_numpy(signals(b, n, seed=1)).cuda()
        st, _, _ = noalloc_denoise(ctx, x_t, "db20", levels, 0, -1.0, True)
        assert st == N.VW_E_WORKSPACE and "intermediate plane" in ctx.last_error()
