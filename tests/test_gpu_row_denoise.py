"""The row-resident SWT denoise kernels against the oracle, bit for bit.

VectorWaveSwtAdapter.denoise (swt/VectorWaveSwtAdapter.java:296-312) on rows short enough for a whole decomposition to
stay on chip runs as ONE launch: detail bands in registers (rows <= 512 samples, <= 4 levels) or in LDS (longer rows /
deeper cascades), exact median in registers, halos that materialise the boundary rule.  Each variant -- and the
materialising three-launch path they replace -- must give the oracle's bits for every threshold rule, boundary mode,
row-length class (even, odd, shorter than a wave's 128 samples, not a multiple of 128) and filter length."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import oracle as o
from util import P, Z, assert_bits_equal, dev_denoise, gpu_denoise, signals
from vectorwave_amd import _native as N
from vectorwave_amd.wavelets import get_wavelet

pytestmark = pytest.mark.gpu

CASES = [  # wavelet, n, levels, rows
    ("db4", 480, 4, 37), ("db4", 481, 4, 5), ("db4", 64, 3, 9), ("db4", 57, 3, 4), ("db4", 512, 4, 3), ("db4", 300, 1, 3),
    ("db4", 513, 4, 3), ("db4", 480, 5, 3), ("db4", 1000, 6, 2), ("db4", 999, 2, 2), ("db4", 2048, 2, 2),
    ("haar", 480, 4, 6), ("haar", 3, 1, 5), ("haar", 17, 4, 5), ("haar", 130, 7, 3), ("haar", 777, 9, 2),
    ("db2", 480, 4, 4), ("db2", 11, 2, 4), ("db2", 255, 6, 2), ("sym4", 480, 4, 3), ("bior2.2", 480, 4, 3), ("db6", 480, 4, 3),
]


@pytest.mark.parametrize("variant", ["regs", "lds", "mat"])
@pytest.mark.parametrize("mode", [P, Z])
@pytest.mark.parametrize("wn,n,j,b", CASES)
def test_row_denoise_bit_exact(ctx, plan, wn, n, j, b, mode, variant):
    if variant == "lds":
        plan("no_band_regs", 1)
    elif variant == "mat":
        plan("no_row_denoise", 1)
    w = get_wavelet(wn)
    if j > o.max_levels(n, len(w.dec_lo)):
        pytest.skip("more levels than the reference allows for this row length")
    x = signals(b, n, seed=n + 7 * j, )
    x[0, : min(n, 5)] = 0.0  # exact zeros and a run of equal magnitudes for the order statistics
    for thr, soft in ((-1.0, True), (-1.0, False), (0.05, True), (0.3, False), (0.0, True)):
        st, want = o.swt_denoise_batch(x, w, j, thr, soft, mode)
        assert st == 0
        st, got, gthr = gpu_denoise(ctx, x, wn, j, mode, thr, soft)
        assert st == N.VW_OK, ctx.last_error()
        assert_bits_equal(got, want, "%s %s N=%d J=%d mode=%d thr=%r soft=%s" % (variant, wn, n, j, mode, thr, soft))
        if thr >= 0:
            assert np.all(gthr == thr)
        else:
            d1 = o.decompose_batch(x, w, j, mode)[1][:, 0, :]
            want_thr = np.array([o.universal_threshold(r) for r in d1])
            assert_bits_equal(gthr, want_thr, "%s thresholds" % variant)


def test_row_denoise_in_place_and_device_pointers(ctx):
    """x and y may be the same device buffer; thresholds_out may be NULL."""
    import torch
    w = get_wavelet("db4")
    x = signals(33, 480, seed=5)
    st, want = o.swt_denoise_batch(x, w, 4, -1.0, True, P)
    x_t = torch.from_numpy(x).cuda()
    out, _ = dev_denoise(ctx, x_t, "db4", 4, P, -1.0, True, out=x_t)
    assert out.data_ptr() == x_t.data_ptr()
    assert_bits_equal(x_t.cpu().numpy(), want, "in place")


N1_CASES = [  # wavelet, n, levels (0 = the single-level form), rows
    ("db4", 480, 4, 9), ("db4", 480, 0, 5), ("db4", 481, 3, 3), ("db4", 64, 2, 4), ("db4", 1000, 5, 2), ("db4", 2048, 2, 2),
    ("haar", 33, 3, 3), ("haar", 480, 4, 3), ("db2", 300, 0, 4), ("sym4", 512, 4, 2), ("db4", 40, 1, 3), ("haar", 64, 0, 2),
]


@pytest.mark.parametrize("variant", ["regs", "lds", "mat"])
@pytest.mark.parametrize("mode", [P, Z])
@pytest.mark.parametrize("wn,n,levels,b", N1_CASES)
def test_wavelet_denoiser_rows_bit_exact(ctx, plan, wn, n, levels, b, mode, variant):
    """WaveletDenoiser on short rows (denoising/WaveletDenoiser.java:121-167, 350-360): UNIVERSAL and MINIMAX (level-scaled
    thresholds from sigma = MAD/0.6745) and the fixed form run in the single-launch row kernels; thresholds per (row, level)
    and denoised rows must equal the oracle's and the three-launch path's bits."""
    from vectorwave_amd import BoundaryMode, ThresholdType, WaveletDenoiser
    if variant == "lds":
        plan("no_band_regs", 1)
    elif variant == "mat":
        plan("no_row_denoise", 1)
    w = get_wavelet(wn)
    x = signals(b, n, seed=3 * n + levels)
    den = WaveletDenoiser(w, BoundaryMode(mode), ctx)
    for method in (o.UNIVERSAL, o.MINIMAX):
        for soft in (True, False):
            st, want, want_thr = o.wavelet_denoise_batch(x, w, levels, method, soft, mode)
            assert st == 0
            got, thr = den.denoise_block(x, levels, method, ThresholdType.SOFT if soft else ThresholdType.HARD,
                                         return_thresholds=True)
            what = "%s %s N=%d J=%d mode=%d method=%d soft=%s" % (variant, wn, n, levels, mode, method, soft)
            assert_bits_equal(thr, want_thr, "thresholds " + what)
            assert_bits_equal(got, want, "denoised " + what)
            if variant != "mat" and len(w.dec_lo) in (2, 4, 8) and n <= 1024:  # longer rows: by LDS need
                assert "k_swt_denoise_row" in ctx.last_plan(), what
    if levels == 0:
        for thr_fixed in (0.25, 0.0):
            for soft in (True, False):
                st, want, _ = o.wavelet_denoise_batch(x, w, 0, o.FIXED, soft, mode, fixed_threshold=thr_fixed)
                assert st == 0
                got = den.denoise_block(x, 0, o.FIXED, ThresholdType.SOFT if soft else ThresholdType.HARD, fixed_threshold=thr_fixed)
                assert_bits_equal(got, want, "%s fixed %r soft=%s" % (variant, thr_fixed, soft))
        # a negative fixed threshold keeps the three-launch path (the row kernels read "negative" as "derive it")
        st, want, _ = o.wavelet_denoise_batch(x, w, 0, o.FIXED, True, mode, fixed_threshold=-0.1)
        got = den.denoise_block(x, 0, o.FIXED, ThresholdType.SOFT, fixed_threshold=-0.1)
        assert_bits_equal(got, want, "%s negative fixed threshold" % variant)


@pytest.mark.parametrize("variant", ["regs", "lds", "mat"])
@pytest.mark.parametrize("mode", [P, Z])
@pytest.mark.parametrize("wn", ["db4", "haar", "db2"])
def test_row_denoise_special_values(ctx, plan, wn, mode, variant):
    """Rows that steer the single-launch kernels between their short thresholding form (finite, moderate samples) and
    the general one: all zeros, negative zeros, subnormals, values beyond 2^500 that stay finite, thresholds that
    shrink every coefficient to zero, a threshold of exactly zero -- zero signs included, every output bit must be
    the oracle's (swt/VectorWaveSwtAdapter.java:255-312)."""
    if variant == "lds":
        plan("no_band_regs", 1)
    elif variant == "mat":
        plan("no_row_denoise", 1)
    w = get_wavelet(wn)
    n, j = 480, 4
    x = signals(12, n, seed=91)
    x[0] = 0.0
    x[1] = -0.0
    x[2] = np.where(np.arange(n) % 3 == 0, -0.0, 0.0)
    x[3] *= 5e-324 / np.abs(x[3]).max() * 40  # subnormals of a few dozen units
    x[4] *= 1e-310
    x[5] *= 1e200     # beyond the short form's bound, finite throughout
    x[6] *= 1e-200
    x[7, ::2] = 0.0   # half of the finest details are small
    x[8] = 1.0        # constant row: every detail coefficient is a rounding residue
    x[9] = -np.abs(x[9])
    x[10, 100] = 3e180
    for thr, soft in ((-1.0, True), (-1.0, False), (0.0, True), (0.0, False), (1e300, True), (1e300, False), (2.5, True)):
        st, want = o.swt_denoise_batch(x, w, j, thr, soft, mode)
        assert st == 0
        st, got, _ = gpu_denoise(ctx, x, wn, j, mode, thr, soft)
        assert st == N.VW_OK, ctx.last_error()
        assert_bits_equal(got, want, "%s %s mode=%d thr=%r soft=%s" % (variant, wn, mode, thr, soft))
