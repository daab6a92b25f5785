"""k_inv_wave6 -- the inverse cascade with one wavefront per run of 512-sample chunks (levels 6..1, filters of 2, 4
or 8 taps, PERIODIC / ZERO_PADDING) -- against the CPU oracle, bit for bit.  The planner takes this kernel by itself
only for large batches; the plan option `inv_wave` = N forces it with runs of N chunks, which is how small inputs
reach every part of it here: whole-row runs, rows split into several runs (warm-up chunks inside the row), rows whose
last chunk is partial (wrap / zeros lane by lane), rows of exactly two chunks.
Reference semantics: modwt/MultiLevelMODWTTransform.java:362-372, 621-660 (reconstruct, applyScaledInverseMODWT)."""
import numpy as np
import pytest

from oracle import oracle as o
from util import assert_bits_equal, gpu_inverse, signals
from vectorwave_amd import _native as N
from vectorwave_amd.wavelets import get_wavelet

pytestmark = pytest.mark.gpu

P, Z = 0, 1


def _inverse_case(ctx, wn, n, j, b, mode, flags=0, seed=None):
    w = get_wavelet(wn)
    x = signals(b, n, seed=(n + j if seed is None else seed))
    st, coeffs = o.decompose_batch(x, w, j, mode)
    assert st == 0
    want = o.reconstruct_batch(coeffs, w, mode)
    st, got, _ = gpu_inverse(ctx, coeffs, wn, mode, flags)
    assert st == 0, ctx.last_error()
    return got, want, coeffs


@pytest.mark.parametrize("mode", [P, Z])
@pytest.mark.parametrize("nch", [1, 2, 3, 5, 1000])
@pytest.mark.parametrize("wn,n,b", [("db4", 16384, 3), ("db4", 1024, 2), ("db4", 1026, 2), ("db4", 5000, 3), ("db4", 4096 + 510, 2),
                                    ("haar", 8192, 2), ("haar", 3074, 3), ("db2", 12290, 2), ("sym4", 4096, 3),
                                    ("db2", 2048, 5)])
def test_wave_inverse_bit_exact(ctx, plan, wn, n, b, nch, mode):
    plan("inv_wave", nch)
    got, want, _ = _inverse_case(ctx, wn, n, 6, b, mode)
    assert "k_inv_wave" in ctx.last_plan(), ctx.last_plan()
    assert_bits_equal(got, want, "wave inverse %s N=%d runs of %d mode %d" % (wn, n, nch, mode))


@pytest.mark.parametrize("mode", [P, Z])
def test_wave_inverse_many_rows(ctx, plan, mode):
    """More rows than the chip holds waves at one time (five per CU): workgroups are dispatched into LDS that earlier
    ones have left, with whatever those left in the head storage."""
    plan("inv_wave", 4)
    got, want, _ = _inverse_case(ctx, "db4", 4096, 6, 1500, mode, seed=7)
    assert "k_inv_wave" in ctx.last_plan(), ctx.last_plan()
    assert_bits_equal(got, want, "wave inverse, 1500 rows x 2 runs")


def test_wave_inverse_fma_mode_within_tolerance(ctx, plan):
    """VW_FLAG_FMA through the same kernel: not bit-identical by design, within the north-star tolerance
    (1e-12 relative per coefficient, here against the largest magnitude of the row)."""
    plan("inv_wave", 1000)
    got, want, _ = _inverse_case(ctx, "db4", 8192, 6, 4, P, flags=N.VW_FLAG_FMA)
    assert "k_inv_wave" in ctx.last_plan(), ctx.last_plan()
    scale = np.abs(want).max(axis=1, keepdims=True)
    assert (np.abs(got - want) <= 1e-12 * scale).all()


def test_wave_inverse_not_taken_where_it_does_not_apply(ctx, plan):
    """Other level groups, thresholding, band masks and finite checks stay with the tile kernels even when forced."""
    plan("inv_wave", 8)
    got, want, coeffs = _inverse_case(ctx, "db4", 8192, 5, 2, P)
    assert "k_inv_wave" not in ctx.last_plan()
    assert_bits_equal(got, want, "five levels")
    got, want, coeffs = _inverse_case(ctx, "db4", 8192, 6, 2, P)
    st, got2, bad = gpu_inverse(ctx, coeffs, "db4", P, N.VW_FLAG_CHECK_FINITE)
    assert st == 0 and bad == -1
    assert "k_inv_wave" not in ctx.last_plan()
    assert_bits_equal(got2, want, "checked inverse")
    plan("inv_wave", -1)
    st, got3, _ = gpu_inverse(ctx, coeffs, "db4", P)
    assert st == 0 and "k_inv_wave" not in ctx.last_plan()
    assert_bits_equal(got3, want, "switched off")


@pytest.mark.parametrize("mode", [P, Z])
@pytest.mark.parametrize("nch", [2, 1000])
@pytest.mark.parametrize("thr,soft", [(-1.0, True), (-1.0, False), (0.25, True), (0.0, True), (float("nan"), True),
                                      (float("nan"), False), (float("inf"), True), (float("inf"), False)])
def test_wave_swt_denoise_bit_exact(ctx, plan, thr, soft, nch, mode):
    """VectorWaveSwtAdapter.denoise at six levels through the wave kernel: the detail bands are thresholded as they land
    (MutableMultiLevelMODWTResult.applyThreshold, modwt/MutableMultiLevelMODWTResult.java:97-114), including thresholds
    that are no finite non-negative number (swt/VectorWaveSwtAdapter.java:296-312: `NaN < 0` is false, so a NaN is
    used as a fixed threshold)."""
    from util import gpu_denoise
    plan("inv_wave", nch)
    w = get_wavelet("db4")
    x = signals(3, 5000, seed=17)
    st, want = o.swt_denoise_batch(x, w, 6, thr, soft, mode)
    assert st == 0
    st, got, _ = gpu_denoise(ctx, x, "db4", 6, mode, thr, soft)
    assert st == N.VW_OK, ctx.last_error()
    assert "k_inv_wave" in ctx.last_plan(), ctx.last_plan()
    assert_bits_equal(got, want, "swt denoise through the wave kernel, threshold %r soft=%s" % (thr, soft))


@pytest.mark.parametrize("method", ["UNIVERSAL", "MINIMAX", "BAYES"])
@pytest.mark.parametrize("soft", [True, False])
def test_wave_wavelet_denoiser_bit_exact(ctx, plan, method, soft):
    """WaveletDenoiser.denoiseMultiLevel (denoising/WaveletDenoiser.java:152-227): per-level thresholds, the soft rule in
    its NaN-propagating form, with a NaN planted in one row's input."""
    import torch
    from util import dev_wavelet_denoise
    plan("inv_wave", 3)
    w = get_wavelet("sym4")
    x = signals(3, 4096, seed=23)
    m = getattr(o, method)
    st, want, wthr = o.wavelet_denoise_batch(x, w, 6, m, soft, 0)
    assert st == 0
    got, thr = dev_wavelet_denoise(ctx, torch.from_numpy(x).cuda(), "sym4", 6, 0, getattr(N, "VW_THRESHOLD_" + method), soft)
    assert "k_inv_wave" in ctx.last_plan(), ctx.last_plan()
    assert_bits_equal(thr.cpu().numpy(), wthr, "thresholds")
    assert_bits_equal(got.cpu().numpy(), want, "WaveletDenoiser %s through the wave kernel" % method)


@pytest.mark.parametrize("mode", [P, Z])
@pytest.mark.parametrize("n", [2050, 2178, 2306, 2434, 3074])
def test_wave_partial_last_chunk_under_load(ctx, plan, n, mode):
    """Rows whose last chunk holds 2, 130, 258 or 386 samples: the chunk issues one to four store instructions, and the
    counted waits of the chunk after it allow for exactly that many.  (A wait that allowed for four let the next chunk
    start before its first bands had landed -- rarely: 600 rows keep every CU busy with waves in different phases.)"""
    plan("inv_wave", 3)
    got, want, _ = _inverse_case(ctx, "db4", n, 6, 600, mode, seed=n)
    assert "k_inv_wave" in ctx.last_plan(), ctx.last_plan()
    assert_bits_equal(got, want, "wave inverse, partial last chunk, N=%d mode %d" % (n, mode))
