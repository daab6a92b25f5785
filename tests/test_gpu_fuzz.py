"""Seeded random sweep over wavelet x length x depth x boundary x batch for the three C-ABI families
(forward/inverse, SWT denoise, WaveletDenoiser): the HIP path against the oracle, bit for bit.  The fixed
parity cases pin the shapes that matter; this sweep looks for the shapes nobody thought of (ragged lengths
around tile and chunk edges, single rows, depth limits, filters longer than the signal).
VW_FUZZ_SEED / VW_FUZZ_SCALE in the environment shift the seeds and multiply the case counts for a longer campaign,
VW_FUZZ_NMAX / VW_FUZZ_BMAX raise the longest random row (default 9000) and the largest batch (default 4); cases whose
oracle would take too long (dense filters of deep levels on long rows) are redrawn with fewer levels.  The test suite
runs the default: 260 cases."""
import os

import numpy as np
import pytest

from oracle import oracle as o
from vectorwave_amd import _native as N
from vectorwave_amd.wavelets import get_wavelet

from util import assert_bits_equal, gpu_denoise, gpu_forward, gpu_inverse

pytestmark = pytest.mark.gpu

SEED_SHIFT = int(os.environ.get("VW_FUZZ_SEED", "0"))
SCALE = max(1, int(os.environ.get("VW_FUZZ_SCALE", "1")))
NMAX = max(2, int(os.environ.get("VW_FUZZ_NMAX", "9000")))
BMAX = max(1, int(os.environ.get("VW_FUZZ_BMAX", "4")))
ORACLE_BUDGET = 1.5e9  # multiply-adds of the reference's dense inverse per case
# VW_FUZZ_PLANS=1: every case also draws one or two plan options (vw_set_plan_option: equivalent launch plans, same bits)
PLANS = os.environ.get("VW_FUZZ_PLANS", "0") != "0"
# VW_FUZZ_DEVICE=1: the denoise cases also go through the device-pointer entry point and the allocation-free one
DEVICE = os.environ.get("VW_FUZZ_DEVICE", "0") != "0"
PLAN_CHOICES = [("no_sym_tiles", [1]), ("no_ring", [1]), ("no_blk", [1]), ("no_sdry", [1]), ("no_row_denoise", [1]), ("no_band_regs", [1]),
                ("no_reg_select", [1]), ("no_wg_select", [1]), ("no_vec2", [1]), ("no_unroll", [1]), ("stream_wgs", [1, 6, 24]),
                ("fwd_chunks", [1, 2, 3, 5, 16]), ("sdry_chunks", [1, 3, 1000]), ("carry_chunks", [2, 3, 8]),
                ("inv_wave", [1, 2, 3, 5, 1000]), ("no_wide_inv", [1]), ("nt_stores", [0, 1]), ("no_soa_runs", [1]),
                ("soa_run_len", [8, 16, 32])]


def _draw_plan(plan, seed):
    if not PLANS:
        return ""
    rng = np.random.default_rng(seed ^ 0x5eed)
    picked = []
    for _ in range(int(rng.integers(1, 3))):
        name, values = PLAN_CHOICES[rng.integers(len(PLAN_CHOICES))]
        value = values[rng.integers(len(values))]
        plan(name, value)
        picked.append("%s=%d" % (name, value))
    return " [" + ",".join(picked) + "]"

WAVELETS = ["haar", "db2", "sym3", "db4", "db6", "db10", "sym4", "sym8", "sym11", "coif1", "coif3", "bior2.2", "bior4.4", "bior3.5", "rbio1.3", "db20"]
EDGES = [1, 2, 3, 7, 8, 9, 31, 63, 64, 65, 255, 256, 257, 479, 480, 481, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049,
         4095, 4096, 4097, 6143, 6145, 8191, 8193]


def _cases(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        wn = WAVELETS[rng.integers(len(WAVELETS))]
        n = int(EDGES[rng.integers(len(EDGES))] if rng.random() < 0.6 else rng.integers(1, NMAX))
        mode = int(rng.integers(3))
        b = int(rng.integers(1, BMAX + 1))
        taps = len(get_wavelet(wn).dec_lo)
        jmax = o.max_levels(n, taps)
        j = int(rng.integers(1, max(jmax, 1) + 1))
        while (NMAX > 9000 or BMAX > 4) and j > 1 and float(n) * b * taps * (1 << j) > ORACLE_BUDGET:
            j -= 1
        out.append((wn, n, j, b, mode, int(rng.integers(1 << 30))))
    return out


@pytest.mark.parametrize("wn,n,j,b,mode,seed", _cases(20250101 + SEED_SHIFT, 120 * SCALE))
def test_random_forward_inverse(ctx, plan, wn, n, j, b, mode, seed):
    w = get_wavelet(wn)
    opts = _draw_plan(plan, seed)
    x = np.ascontiguousarray(np.random.default_rng(seed).standard_normal((b, n)))
    st, want = o.decompose_batch(x, w, j, mode)
    gst, got, _ = gpu_forward(ctx, x, wn, j, mode)
    if st != 0:
        assert gst == N.VW_E_LEVEL  # beyond the reference's level limit
        return
    assert gst == 0, ctx.last_error()
    assert_bits_equal(got, want, "forward %s N=%d J=%d mode=%d%s" % (wn, n, j, mode, opts))
    gst, gotx, _ = gpu_inverse(ctx, want, wn, mode)
    if len(w.rec_hi) < len(w.rec_lo) and mode != 2:
        assert gst == N.VW_E_LENGTH_MISMATCH  # the reference indexes out of bounds here
        return
    assert gst == 0, ctx.last_error()
    assert_bits_equal(gotx, o.reconstruct_batch(want, w, mode), "inverse %s N=%d J=%d mode=%d%s" % (wn, n, j, mode, opts))


@pytest.mark.parametrize("wn,n,j,b,mode,seed", _cases(20250202 + SEED_SHIFT, 60 * SCALE))
def test_random_denoise(ctx, plan, wn, n, j, b, mode, seed):
    w = get_wavelet(wn)
    if len(w.rec_hi) < len(w.rec_lo) and mode != 2:
        pytest.skip("the reference cannot invert this wavelet")
    opts = _draw_plan(plan, seed)
    rng = np.random.default_rng(seed)
    x = np.ascontiguousarray(np.sin(np.arange(n) / 5.0)[None] + 0.3 * rng.standard_normal((b, n)))
    soft = bool(seed & 1)
    thr = -1.0 if seed & 2 else 0.2
    st, want = o.swt_denoise_batch(x, w, j, thr, soft, mode)
    gst, got, _ = gpu_denoise(ctx, x, wn, j, mode, thr, soft)
    if st != 0:
        assert gst == N.VW_E_LEVEL
        return
    assert gst == 0, ctx.last_error()
    assert_bits_equal(got, want, "swt denoise %s N=%d J=%d mode=%d%s" % (wn, n, j, mode, opts))
    if DEVICE:
        import torch
        from test_zz_no_alloc import noalloc_denoise
        from util import dev_denoise
        x_t = torch.from_numpy(x).cuda()
        dgot, _ = dev_denoise(ctx, x_t, wn, j, mode, thr, soft)
        assert_bits_equal(dgot.cpu().numpy(), want, "device entry point %s N=%d J=%d mode=%d%s" % (wn, n, j, mode, opts))
        nst, ngot, _ = noalloc_denoise(ctx, x_t, wn, j, mode, thr, soft)
        assert nst == 0, ctx.last_error()
        assert_bits_equal(ngot.cpu().numpy(), want, "allocation-free entry point %s N=%d J=%d mode=%d%s" % (wn, n, j, mode, opts))
    method = [o.UNIVERSAL, o.MINIMAX, o.BAYES][seed % 3]
    if n > (2000 if NMAX <= 9000 else NMAX):
        return
    st, wantd, wthr = o.wavelet_denoise_batch(x, w, j, method, soft, mode)
    from vectorwave_amd import BoundaryMode, ThresholdType, WaveletDenoiser
    gotd, gthr = WaveletDenoiser(w, BoundaryMode(mode), ctx).denoise_block(x, j, method, ThresholdType.SOFT if soft else ThresholdType.HARD,
                                                                         return_thresholds=True)
    assert_bits_equal(gthr, wthr, "WaveletDenoiser thresholds %s N=%d J=%d method=%d" % (wn, n, j, method))
    assert_bits_equal(gotd, wantd, "WaveletDenoiser %s N=%d J=%d method=%d" % (wn, n, j, method))


@pytest.mark.parametrize("wn,n,j,b,mode,seed", _cases(20250404 + SEED_SHIFT, 80 * SCALE))
def test_random_band_masks_and_finite_checks(ctx, wn, n, j, b, mode, seed):
    """The band-selective inverse (reconstructFromLevel / reconstructLevels, modwt/MultiLevelMODWTTransform.java:385-468)
    with a random mask, and the finite checks of both directions (util/ValidationUtils.java:106-117: the first
    non-finite flat index) with one or two non-finite values planted at random places."""
    import ctypes as C
    from util import _dp, _t, _vp
    w = get_wavelet(wn)
    rng = np.random.default_rng(seed)
    x = np.ascontiguousarray(rng.standard_normal((b, n)))
    st, c = o.decompose_batch(x, w, j, mode)
    if st != 0:
        return
    if len(w.rec_hi) < len(w.rec_lo) and mode != 2:
        return
    hr, gr = _t(w.rec_lo), _t(w.rec_hi)
    mask = int(rng.integers(1, 1 << (j + 1)))
    z = c.copy()
    for k in range(j + 1):
        if not (mask >> k) & 1:
            z[:, k, :] = 0.0
    out = np.full((b, n), np.nan)
    gst = ctx.lib.vw_modwt_inverse_bands_batch(ctx.handle, _vp(c), b, n, j, _dp(hr), hr.size, _dp(gr), gr.size, mode, mask, _vp(out))
    assert gst == 0, ctx.last_error()
    assert_bits_equal(out, o.reconstruct_batch(z, w, mode), "masked inverse %s N=%d J=%d mode=%d mask=%#x" % (wn, n, j, mode, mask))
    # finite checks: clean input passes with the same bits, a planted NaN / inf is reported at its flat index
    gst, got, bad = gpu_forward(ctx, x, wn, j, mode, N.VW_FLAG_CHECK_FINITE)
    assert gst == 0 and bad == -1, ctx.last_error()
    assert_bits_equal(got, c, "checked forward %s N=%d J=%d mode=%d" % (wn, n, j, mode))
    xb = x.copy()
    at = sorted(int(v) for v in rng.integers(0, b * n, size=2))
    xb.reshape(-1)[at[0]] = np.nan
    xb.reshape(-1)[at[1]] = np.inf
    gst, _, bad = gpu_forward(ctx, xb, wn, j, mode, N.VW_FLAG_CHECK_FINITE)
    assert gst == N.VW_E_NONFINITE and bad == at[0], (gst, bad, at)
    gst, gotx, bad = gpu_inverse(ctx, c, wn, mode, N.VW_FLAG_CHECK_FINITE)
    assert gst == 0 and bad == -1, ctx.last_error()
    assert_bits_equal(gotx, o.reconstruct_batch(c, w, mode), "checked inverse %s N=%d J=%d mode=%d" % (wn, n, j, mode))
    cb = c.copy()
    at = sorted(int(v) for v in rng.integers(0, cb.size, size=2))
    cb.reshape(-1)[at[0]] = np.inf
    cb.reshape(-1)[at[1]] = np.nan
    gst, _, bad = gpu_inverse(ctx, cb, wn, mode, N.VW_FLAG_CHECK_FINITE)
    assert gst == N.VW_E_NONFINITE and bad == at[0], (gst, bad, at)


def _single_cases(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        wn = WAVELETS[rng.integers(len(WAVELETS))]
        n = int(EDGES[rng.integers(len(EDGES))] if rng.random() < 0.5 else rng.integers(1, NMAX))
        out.append((wn, n, int(rng.integers(1, 4 * BMAX + 1)), int(rng.integers(3)), int(rng.integers(1 << 30))))
    return out


@pytest.mark.parametrize("wn,n,b,mode,seed", _single_cases(20250505 + SEED_SHIFT, 80 * SCALE))
def test_random_single_level_and_soa(ctx, plan, wn, n, b, mode, seed):
    """MODWTTransform.forward / inverse / inverseBatchOptimized (modwt/MODWTTransform.java:193-287, 600-660) and
    BatchSIMDMODWT.batchMODWTSoA (modwt/BatchSIMDMODWT.java:64-274) on random shapes."""
    w = get_wavelet(wn)
    opts = _draw_plan(plan, seed)
    x = np.ascontiguousarray(np.random.default_rng(seed).standard_normal((b, n)))
    want = o.modwt_forward_batch(x, w, mode)
    st, got, _ = gpu_forward(ctx, x, wn, 1, mode, N.VW_FLAG_SINGLE_LEVEL)
    assert st == 0, ctx.last_error()
    assert_bits_equal(got, want, "single forward %s N=%d b=%d mode=%d%s" % (wn, n, b, mode, opts))
    if len(w.rec_hi) >= len(w.rec_lo):
        for variant, flag in ((0, 0), (1, N.VW_FLAG_SYMMETRIC_BATCH_VARIANT)):
            wantx = o.modwt_inverse_batch(want, w, mode, batch_variant=variant)
            st, gotx, _ = gpu_inverse(ctx, want, wn, mode, N.VW_FLAG_SINGLE_LEVEL | flag)
            assert st == 0, ctx.last_error()
            assert_bits_equal(gotx, wantx, "single inverse variant %d %s N=%d b=%d mode=%d%s" % (variant, wn, n, b, mode, opts))
    taps = len(w.dec_lo)
    if len(w.dec_hi) == taps and n >= taps - 1:  # (the reference's index (t - l + N) % N goes negative below that)
        from vectorwave_amd import BatchSIMDMODWT
        soa = o.convert_to_soa(x)
        a, d = np.empty(b * n), np.empty(b * n)
        BatchSIMDMODWT.batchMODWTSoA(soa, a, d, w, b, n, ctx)
        wa, wd = o.batch_modwt_soa(soa, w, b, n)
        assert_bits_equal(a, wa, "soaApprox %s %d x %d" % (wn, b, n))
        assert_bits_equal(d, wd, "soaDetail %s %d x %d" % (wn, b, n))


def _stream_cases(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        wn = ["haar", "db2", "db4", "sym8", "coif1"][rng.integers(5)]
        n = int([1, 2, 3, 8, 31, 32, 33, 64, 65, 100, 255, 256, 480, 1000, 2049][rng.integers(15)])
        window = int([1, 2, 7, 10, 11, 63, 64, 100, 256, 1000, 1024, 2048, 2500][rng.integers(13)])
        est = int(rng.integers(3))
        mult = [1.0, 1.0, 0.5, 1.5, 3.0][rng.integers(5)]
        method = int([0, 1, 2, 3, 0, 1, 2, 4][rng.integers(8)])
        if mult == 1.0 and method == 4:
            method = 0
        if mult != 1.0 and method == 3:
            method = 2
        out.append((wn, n, window, est, mult, method, int(rng.integers(3)), bool(rng.integers(2)), int(rng.integers(1 << 30))))
    return out


@pytest.mark.parametrize("wn,n,window,est,mult,method,mode,soft,seed", _stream_cases(20250303 + SEED_SHIFT, 80 * SCALE))
def test_random_streaming_denoiser(ctx, wn, n, window, est, mult, method, mode, soft, seed):
    from vectorwave_amd import (BoundaryMode, MODWTStreamingDenoiser, NoiseEstimation, ThresholdMethod, ThresholdType)
    w = get_wavelet(wn)
    rng = np.random.default_rng(seed)
    streams = 3
    state = o.StreamingDenoiserState(streams, window)
    den = (MODWTStreamingDenoiser.Builder().wavelet(w).boundaryMode(BoundaryMode(mode)).noiseWindowSize(window)
           .thresholdMethod(ThresholdMethod(method)).thresholdType(ThresholdType.SOFT if soft else ThresholdType.HARD)
           .thresholdMultiplier(mult).noiseEstimation(NoiseEstimation(est)).streams(streams).context(ctx).build())
    with den:
        for k in range(4):
            blk = np.ascontiguousarray(np.sin(np.arange(n) / 3.0 + k)[None] + 0.2 * rng.standard_normal((streams, n)))
            st, want, _ = o.streaming_denoise_batch(state, blk, w, method, soft, mult, est, mode)
            assert st == 0
            tag = "%s n=%d ring=%d est=%d mult=%g method=%d mode=%d block %d" % (wn, n, window, est, mult, method, mode, k)
            assert_bits_equal(den.denoise(blk), want, tag)
            win, idx = den.getNoiseWindow()
            assert idx == state.index
            assert_bits_equal(win, state.window, "ring " + tag)
            assert_bits_equal(den.getEstimatedNoiseLevel(), state.level, "level " + tag)
