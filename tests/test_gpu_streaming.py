"""GPU parity of the streaming denoiser (modwt/streaming/MODWTStreamingDenoiser.java) against the oracle over
sequences of blocks: denoised blocks, thresholds, noise rings, ring position and noise levels bit for bit, for
every noise estimator, threshold rule, multiplier, boundary mode and ring/block length relation; plus the
Python mirror's life cycle and error behaviour."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import oracle as o
from vectorwave_amd import _native as N
from vectorwave_amd import (BoundaryMode, IllegalStateException, InvalidArgumentException, InvalidSignalException,
                            MODWTStreamingDenoiser, NoiseEstimation, ThresholdMethod, ThresholdType)
from vectorwave_amd.wavelets import get_wavelet

from util import P, S, Z, assert_bits_equal

pytestmark = pytest.mark.gpu


def _blocks(streams, n, count, seed, noise=0.1):
    rng = np.random.default_rng(seed)
    t = np.arange(count * n).reshape(count, n)
    clean = np.sin(2 * np.pi * t / 32.0)
    return np.ascontiguousarray(clean[:, None, :] + noise * rng.standard_normal((count, streams, n)))


def _build(ctx, wn, streams, window, mode=P, method=o.UNIVERSAL, soft=True, mult=1.0, est=o.NOISE_MAD):
    return (MODWTStreamingDenoiser.Builder().wavelet(wn).boundaryMode(BoundaryMode(mode)).noiseWindowSize(window)
            .thresholdMethod(ThresholdMethod(method)).thresholdType(ThresholdType.SOFT if soft else ThresholdType.HARD)
            .thresholdMultiplier(mult).noiseEstimation(NoiseEstimation(est)).streams(streams).context(ctx).build())


def _run_and_compare(ctx, wn, streams, n, window, count, mode, method, soft, mult, est, seed):
    w = get_wavelet(wn)
    state = o.StreamingDenoiserState(streams, window)
    tag = "%s n=%d w=%d mode=%d method=%d mult=%g est=%d" % (wn, n, window, mode, method, mult, est)
    with _build(ctx, wn, streams, window, mode, method, soft, mult, est) as den:
        for k, blk in enumerate(_blocks(streams, n, count, seed)):
            st, want, want_thr = o.streaming_denoise_batch(state, blk, w, method, soft, mult, est, mode)
            assert st == 0
            got = den.denoise(blk)
            assert_bits_equal(got, want, "block %d %s" % (k, tag))
            win, idx = den.getNoiseWindow()
            assert idx == state.index
            assert_bits_equal(win, state.window, "ring after block %d %s" % (k, tag))
            assert_bits_equal(np.atleast_1d(den.getEstimatedNoiseLevel()), state.level, "level after block %d %s" % (k, tag))
        assert den.getSamplesProcessed() == count * n


@pytest.mark.parametrize("est", [o.NOISE_MAD, o.NOISE_STD, o.NOISE_FIXED])
@pytest.mark.parametrize("mult", [1.0, 1.5])
@pytest.mark.parametrize("n,window", [(64, 1024), (256, 1024), (480, 100), (1000, 7), (300, 300), (301, 300), (45, 11), (37, 10),
                                      (2048, 1024), (128, 1)])
def test_streaming_denoiser_bit_exact(ctx, n, window, mult, est):
    _run_and_compare(ctx, "db4", 5, n, window, 4, P, o.UNIVERSAL, True, mult, est, seed=n + window)


@pytest.mark.parametrize("mode", [P, Z, S])
@pytest.mark.parametrize("soft", [True, False])
@pytest.mark.parametrize("method,mult", [(o.UNIVERSAL, 1.0), (o.SURE, 1.0), (o.MINIMAX, 1.0), (o.BAYES, 1.0), (o.UNIVERSAL, 0.7),
                                         (o.SURE, 2.0), (o.MINIMAX, 1.3), (o.FIXED, 2.5)])
def test_streaming_methods_and_boundaries(ctx, method, mult, soft, mode):
    _run_and_compare(ctx, "sym8", 3, 200, 128, 3, mode, method, soft, mult, o.NOISE_MAD, seed=method + mode)


@pytest.mark.parametrize("n", [20, 50, 64, 65])
def test_streaming_minimax_multiplier_pieces(ctx, n):
    _run_and_compare(ctx, "haar", 2, n, 32, 3, P, o.MINIMAX, True, 2.0, o.NOISE_STD, seed=n)


def test_rings_and_blocks_longer_than_the_lds_stage(ctx):
    # ring of 7000 and blocks of 9000 samples: both medians read global memory instead of LDS
    _run_and_compare(ctx, "db4", 3, 9000, 7000, 2, P, o.UNIVERSAL, True, 1.2, o.NOISE_MAD, seed=1)
    _run_and_compare(ctx, "db4", 3, 9000, 7000, 2, P, o.UNIVERSAL, True, 1.2, o.NOISE_STD, seed=2)
    _run_and_compare(ctx, "db4", 2, 7001, 6144, 2, P, o.SURE, False, 0.9, o.NOISE_FIXED, seed=3)


@pytest.mark.parametrize("window", [1, 2, 3, 64, 65, 255, 256, 257, 1024, 1025, 2048, 2049, 5000])
def test_mad_over_awkward_rings(ctx, window):
    # rings of every length class (registers up to 2048 entries, LDS, global) fed blocks whose details repeat,
    # vanish or span the whole exponent range: both medians of the MAD must still be the exact order statistics
    w = get_wavelet("haar")
    rng = np.random.default_rng(window)
    n = 96
    blocks = [np.round(rng.standard_normal((4, n)), 0), np.zeros((4, n)), rng.choice([1e-300, 1.0, 1e300], (4, n)),
              rng.standard_normal((4, n)), np.tile(rng.standard_normal((4, 2)), (1, n // 2)), rng.standard_normal((4, n)) * 1e-160]
    for mult in (1.0, 0.5):
        state = o.StreamingDenoiserState(4, window)
        with _build(ctx, "haar", 4, window, mult=mult) as den:
            for k, blk in enumerate(blocks):
                blk = np.ascontiguousarray(blk)
                st, want, _ = o.streaming_denoise_batch(state, blk, w, o.UNIVERSAL, True, mult, o.NOISE_MAD, P)
                assert st == 0
                assert_bits_equal(den.denoise(blk), want, "block %d ring %d" % (k, window))
                assert_bits_equal(den.getEstimatedNoiseLevel(), state.level, "level after block %d ring %d" % (k, window))


def test_constant_and_zero_blocks(ctx):
    # all-zero details leave MAD and STD at 0 and the multiplier path at threshold 0
    w = get_wavelet("haar")
    for est in (o.NOISE_MAD, o.NOISE_STD):
        state = o.StreamingDenoiserState(2, 16)
        with _build(ctx, "haar", 2, 16, mult=2.0, est=est) as den:
            for blk in (np.zeros((2, 40)), np.full((2, 40), 3.25), _blocks(2, 40, 1, 8)[0]):
                st, want, _ = o.streaming_denoise_batch(state, blk, w, o.UNIVERSAL, True, 2.0, est, P)
                assert st == 0
                assert_bits_equal(den.denoise(blk), want, "constant block")
                assert_bits_equal(np.atleast_1d(den.getEstimatedNoiseLevel()), state.level, "level")


def test_many_streams_headline_shape(ctx):
    # 8192 channels x 480-sample blocks, default ring of 1024: a sample of rows against the oracle
    w = get_wavelet("db4")
    streams, n = 8192, 480
    rows = np.array([0, 1, 4095, 8191])
    state = o.StreamingDenoiserState(rows.size, 1024)
    with _build(ctx, "db4", streams, 1024, mult=1.1) as den:
        for blk in _blocks(streams, n, 3, seed=77):
            st, want, _ = o.streaming_denoise_batch(state, np.ascontiguousarray(blk[rows]), w, o.UNIVERSAL, True, 1.1, o.NOISE_MAD, P)
            got = den.denoise(blk)
            assert_bits_equal(got[rows], want, "sampled rows")
        assert_bits_equal(den.getEstimatedNoiseLevel()[rows], state.level, "levels")


def test_single_stream_mirror_and_subscribers(ctx):
    w = get_wavelet("db4")
    den = MODWTStreamingDenoiser.Builder().context(ctx).build()  # reference defaults: DB4, PERIODIC, ring 1024, MAD
    assert (den.bufferSize, den.noiseWindowSize, den.thresholdMultiplier) == (256, 1024, 1.0)
    seen = []
    den.subscribe(seen.append)
    state = o.StreamingDenoiserState(1, 1024)
    for blk in _blocks(1, 256, 3, seed=5):
        want = o.streaming_denoise_batch(state, blk, w)[1][0]
        got = den.denoise(blk[0])
        assert got.shape == (256,)
        assert_bits_equal(got, want, "single stream")
    assert len(seen) == 3 and seen[-1] is not got and np.array_equal(seen[-1], got)
    assert den.getEstimatedNoiseLevel() == state.level[0]
    assert den.getSamplesProcessed() == 768
    den.close()
    den.close()
    assert den.isClosed() and den.getSamplesProcessed() == 768
    with pytest.raises(IllegalStateException):
        den.denoise(np.zeros(8))


def test_error_behaviour(ctx):
    B = MODWTStreamingDenoiser.Builder
    for bad in (lambda: B().wavelet(None), lambda: B().boundaryMode(None), lambda: B().bufferSize(0),
                lambda: B().thresholdType(None), lambda: B().thresholdMethod(None), lambda: B().thresholdMultiplier(0.0),
                lambda: B().thresholdMultiplier(-1.0), lambda: B().noiseEstimation(None), lambda: B().noiseWindowSize(0)):
        with pytest.raises(InvalidArgumentException):
            bad()
    with B().context(ctx).build() as den:
        with pytest.raises(InvalidArgumentException):
            den.denoise(None)
        with pytest.raises(InvalidArgumentException):
            den.denoise(np.zeros(0))
        with pytest.raises(InvalidSignalException):
            den.denoise(np.array([1.0, np.nan, 2.0, 3.0]))
    # denoise(samples, FIXED, type) throws in WaveletDenoiser; calculateThreshold has no BAYES branch
    with B().thresholdMethod(ThresholdMethod.FIXED).context(ctx).build() as den:
        with pytest.raises(InvalidArgumentException):
            den.denoise(np.arange(16.0))
    with B().thresholdMethod(ThresholdMethod.BAYES).thresholdMultiplier(2.0).context(ctx).build() as den:
        with pytest.raises(InvalidArgumentException):
            den.denoise(np.arange(16.0))


def test_c_abi_argument_checks(ctx):
    w = get_wavelet("db4")
    dp = C.POINTER(C.c_double)
    h, g, hr, gr = (np.ascontiguousarray(t, dtype=np.float64) for t in (w.dec_lo, w.dec_hi, w.rec_lo, w.rec_hi))
    state = C.c_void_p()
    assert ctx.lib.vw_stream_state_create(ctx.handle, 0, 16, C.byref(state)) == N.VW_E_NOT_POSITIVE
    assert ctx.lib.vw_stream_state_create(ctx.handle, 2, 0, C.byref(state)) == N.VW_E_NOT_POSITIVE
    assert ctx.lib.vw_stream_state_create(ctx.handle, 2, 16, C.byref(state)) == N.VW_OK
    x = np.zeros((3, 32))
    y = np.empty_like(x)

    def call(b, mult=1.0, est=0, st=state):
        return ctx.lib.vw_streaming_denoise_batch(ctx.handle, st, C.c_void_p(x.ctypes.data), b, 32, h.ctypes.data_as(dp), h.size,
                                                  g.ctypes.data_as(dp), g.size, hr.ctypes.data_as(dp), hr.size,
                                                  gr.ctypes.data_as(dp), gr.size, 0, 0, 0, 1, mult, est, -1.0, -1.0,
                                                  C.c_void_p(y.ctypes.data), None)
    assert call(3) == N.VW_E_LENGTH_MISMATCH      # a state of 2 streams
    assert call(2, mult=0.0) == N.VW_E_NOT_POSITIVE
    assert call(2, est=3) == N.VW_E_UNSUPPORTED
    assert call(2, st=None) == N.VW_E_NULL
    assert call(2) == N.VW_OK
    assert ctx.lib.vw_stream_state_samples_processed(state) == 32
    assert ctx.lib.vw_stream_state_destroy(ctx.handle, state) == N.VW_OK


@pytest.mark.parametrize("hook", ["no_reg_select", "no_std_chains"])
def test_long_row_kernels_on_short_rows(ctx, plan, hook):
    # the 256-lane LDS kernels normally serve rings and blocks beyond 2048 samples; the hooks route short rows
    # through them too, so their whole logic is exercised on the shapes the oracle checks quickly
    plan(hook, 1)
    for est in (o.NOISE_MAD, o.NOISE_STD):
        for mult in (1.0, 1.5):
            _run_and_compare(ctx, "db4", 3, 300, 128, 3, P, o.UNIVERSAL, True, mult, est, seed=est + 10)
            _run_and_compare(ctx, "db4", 3, 100, 1000, 3, P, o.MINIMAX, False, mult, est, seed=est + 20)
