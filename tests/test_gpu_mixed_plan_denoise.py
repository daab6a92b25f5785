"""Denoise through plans that mix the tile kernels with the generic one-level inverse.

A level whose context fits neither a tile kernel's halo nor the long-context kernel's history (22 taps at level 9:
21 * 256 = 5376 samples) is undone by the generic one-lane-per-sample kernel, between tile-kernel launches for the
other levels.  In a thresholding inverse every launch thresholds the bands it reads; the generic kernel did not, so
W_9 went into the reconstruction unthresholded (found by the scaled-up random sweep: sym11, N = 8191, J = 9).
Reference: swt/VectorWaveSwtAdapter.java:296-312 (every detail level is thresholded),
denoising/WaveletDenoiser.java:152-227."""
import numpy as np
import pytest

from oracle import oracle as o
from util import assert_bits_equal, gpu_denoise, gpu_inverse
from vectorwave_amd.wavelets import get_wavelet

pytestmark = pytest.mark.gpu

P, Z = 0, 1

CASES = [("sym11", 8191, 9, 4), ("sym11", 8192, 9, 3), ("sym11", 6000, 9, 2), ("db10", 6002, 9, 2), ("db20", 6001, 8, 2)]


def _signal(b, n, seed):
    rng = np.random.default_rng(seed)
    return np.ascontiguousarray(np.sin(np.arange(n) / 5.0)[None] + 0.3 * rng.standard_normal((b, n)))


@pytest.mark.parametrize("mode", [P, Z])
@pytest.mark.parametrize("thr,soft", [(0.2, False), (0.2, True), (-1.0, True), (-1.0, False)])
@pytest.mark.parametrize("wn,n,j,b", CASES)
def test_swt_denoise_with_a_generic_level_in_the_plan(ctx, wn, n, j, b, thr, soft, mode):
    w = get_wavelet(wn)
    x = _signal(b, n, n + j)
    st, want = o.swt_denoise_batch(x, w, j, thr, soft, mode)
    assert st == 0
    st, got, _ = gpu_denoise(ctx, x, wn, j, mode, thr, soft)
    assert st == 0, ctx.last_error()
    plan = ctx.last_plan()
    if wn == "sym11":  # (the other shapes are neighbours: deep cascades of long filters whose plans may be all tiles)
        assert "k_level_inv" in plan and "k_inv_fused" in plan, plan
    assert_bits_equal(got, want, "swt denoise %s N=%d J=%d mode=%d thr=%g soft=%d" % (wn, n, j, mode, thr, soft))


@pytest.mark.parametrize("mode", [P, Z])
@pytest.mark.parametrize("method", ["UNIVERSAL", "MINIMAX", "BAYES"])
@pytest.mark.parametrize("wn,n,j,b", CASES[:3])
def test_wavelet_denoiser_with_a_generic_level_in_the_plan(ctx, wn, n, j, b, method, mode):
    """Per-level thresholds (one per row and level) reach the generic kernel with the level's offset."""
    from vectorwave_amd import BoundaryMode, ThresholdType, WaveletDenoiser
    w = get_wavelet(wn)
    x = _signal(b, n, n + 3 * j)
    m = getattr(o, method)
    for soft in (True, False):
        st, want, wthr = o.wavelet_denoise_batch(x, w, j, m, soft, mode)
        assert st == 0
        got, gthr = WaveletDenoiser(w, BoundaryMode(mode), ctx).denoise_block(
            x, j, m, ThresholdType.SOFT if soft else ThresholdType.HARD, return_thresholds=True)
        assert "k_level_inv" in ctx.last_plan(), ctx.last_plan()
        assert_bits_equal(gthr, wthr, "thresholds %s %s" % (wn, method))
        assert_bits_equal(got, want, "WaveletDenoiser %s N=%d J=%d %s soft=%d mode=%d" % (wn, n, j, method, soft, mode))


@pytest.mark.parametrize("wn,n,j,b", CASES)
def test_plain_inverse_of_the_same_plans(ctx, wn, n, j, b):
    w = get_wavelet(wn)
    x = _signal(b, n, n)
    st, c = o.decompose_batch(x, w, j, P)
    assert st == 0
    st, got, _ = gpu_inverse(ctx, c, wn, P)
    assert st == 0, ctx.last_error()
    assert wn != "sym11" or "k_level_inv" in ctx.last_plan(), ctx.last_plan()
    assert_bits_equal(got, o.reconstruct_batch(c, w, P), "inverse %s N=%d J=%d" % (wn, n, j))
