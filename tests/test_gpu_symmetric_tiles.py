"""SYMMETRIC through the tile kernels, bit for bit against the CPU oracle.

Forward: the tile-streaming kernel, whose first tile of a row takes its left halo as the mirror image of the tile's own
first samples after every level (the reference reflects the index into each level's own input,
internal/ScalarOps.java:827-845, util/MathUtils.java:31-52).  Inverse: k_inv_sym, the left-looking tile kernel of the
reference's SYMMETRIC reconstruction (modwt/MultiLevelMODWTTransform.java:664-685: reflect(t - l s), the two products
added one after the other, per-filter lengths) and of MODWTTransform.inverse (modwt/MODWTTransform.java:269-287: pair
sum).  inverseBatchOptimized's flavour (reflects t + l at the row end, :644-657) is the ordinary single-level tile kernel
with a mirrored right halo.  Rows shorter than one chunk and level groups whose context exceeds a chunk stay with the
generic one-level kernels; `no_sym_tiles` forces those everywhere."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as o
from util import _dp, _t, _vp, assert_bits_equal, gpu_denoise, gpu_forward, gpu_inverse, signals
from vectorwave_amd import _native as N
from vectorwave_amd.wavelets import get_wavelet

pytestmark = pytest.mark.gpu

S = 2  # VW_BOUNDARY_SYMMETRIC

# (wavelet, n, levels, rows)
ML = [
    ("db4", 16384, 6, 3),     # the headline shape: one level group, 2048-sample tiles
    ("db4", 512, 6, 2),       # the shortest row the tile path takes: the only tile is the mirrored one
    ("db4", 514, 5, 2),
    ("db4", 1000, 4, 3),      # 512-sample tiles, the last one partial
    ("db4", 2048, 6, 2),
    ("db4", 2050, 6, 2),      # one sample pair into a second tile
    ("db4", 4097, 5, 2),      # odd length: rows not 16-byte aligned, one sample per access
    ("db4", 5000, 6, 3),
    ("haar", 3000, 8, 2),     # context 255: eight levels in one group
    ("db2", 8192, 7, 2),
    ("sym4", 4096, 3, 4),
    ("coif1", 6000, 5, 2),    # six taps: the generic tap loop
    ("db6", 7000, 5, 2),      # context 341
    ("db4", 9000, 7, 2),      # context 889: two level groups, ping-pong through the workspace
    ("sym8", 12000, 6, 2),    # context 945: groups of (1..5) and (6)
    ("bior2.2", 4096, 4, 2),  # filters of different lengths (5 / 3 taps)
    ("rbio2.4", 5000, 3, 2),  # 3 / 9 taps
]


@pytest.mark.parametrize("wn,n,j,b", ML)
def test_symmetric_multilevel_through_tiles(ctx, wn, n, j, b):
    w = get_wavelet(wn)
    x = signals(b, n, seed=n + j)
    st, want = o.decompose_batch(x, w, j, S)
    assert st == 0
    st, got, _ = gpu_forward(ctx, x, wn, j, S)
    assert st == 0, ctx.last_error()
    assert "k_fwd_stream" in ctx.last_plan() and "k_level_fwd" not in ctx.last_plan(), ctx.last_plan()
    assert_bits_equal(got, want, "SYMMETRIC forward %s N=%d J=%d" % (wn, n, j))
    want_x = o.reconstruct_batch(want, w, S)
    st, back, _ = gpu_inverse(ctx, want, wn, S)
    assert st == 0, ctx.last_error()
    assert "k_inv_sym" in ctx.last_plan() and "k_level_inv" not in ctx.last_plan(), ctx.last_plan()
    assert_bits_equal(back, want_x, "SYMMETRIC inverse %s N=%d J=%d" % (wn, n, j))


def test_symmetric_tiles_equal_the_generic_kernels(ctx, plan):
    """The same call with the tile path switched off (the kernels every SYMMETRIC call used to take)."""
    x = signals(5, 6000, seed=3)
    st, tiled, _ = gpu_forward(ctx, x, "db4", 6, S)
    assert st == 0 and "k_fwd_stream" in ctx.last_plan()
    st, tback, _ = gpu_inverse(ctx, tiled, "db4", S)
    assert st == 0 and "k_inv_sym" in ctx.last_plan()
    plan("no_sym_tiles", 1)
    st, generic, _ = gpu_forward(ctx, x, "db4", 6, S)
    assert st == 0 and "k_fwd_stream" not in ctx.last_plan()
    st, gback, _ = gpu_inverse(ctx, generic, "db4", S)
    assert st == 0 and "k_inv_sym" not in ctx.last_plan()
    assert_bits_equal(tiled, generic, "forward")
    assert_bits_equal(tback, gback, "inverse")


@pytest.mark.parametrize("wn,n,b", [("db4", 4096, 6), ("db4", 1000, 5), ("haar", 777, 4), ("bior2.2", 2048, 4), ("db4", 4097, 4),
                                    ("db4", 512, 3), ("sym8", 5000, 3), ("db10", 2049, 2), ("rbio2.4", 3000, 2)])
def test_symmetric_single_level_through_tiles(ctx, wn, n, b):
    """MODWTTransform.forward / inverse (reflects t - l, pair sum) on tiles; the batch flavour of the inverse (t + l) not."""
    w = get_wavelet(wn)
    x = signals(b, n, seed=n)
    want = o.modwt_forward_batch(x, w, S)
    st, got, _ = gpu_forward(ctx, x, wn, 1, S, N.VW_FLAG_SINGLE_LEVEL)
    assert st == 0, ctx.last_error()
    assert "k_fwd_stream" in ctx.last_plan(), ctx.last_plan()
    assert_bits_equal(got, want, "single-level forward")
    if len(w.rec_hi) >= len(w.rec_lo):  # (the reference indexes the high-pass filter with the low-pass loop bound)
        st, back, _ = gpu_inverse(ctx, want, wn, S, N.VW_FLAG_SINGLE_LEVEL)
        assert st == 0, ctx.last_error()
        assert "k_inv_sym" in ctx.last_plan(), ctx.last_plan()
        assert_bits_equal(back, o.modwt_inverse_batch(want, w, S, batch_variant=0), "inverse()")
        st, back2, _ = gpu_inverse(ctx, want, wn, S, N.VW_FLAG_SINGLE_LEVEL | N.VW_FLAG_SYMMETRIC_BATCH_VARIANT)
        assert st == 0, ctx.last_error()
        # (reflects t + l at the row END: the ordinary right-looking tile kernel with a mirrored halo)
        assert ("k_inv_fused" in ctx.last_plan()) == (n >= 512), ctx.last_plan()
        assert_bits_equal(back2, o.modwt_inverse_batch(want, w, S, batch_variant=1), "inverseBatchOptimized()")


def test_symmetric_short_rows_and_deep_contexts_stay_generic(ctx):
    x = signals(3, 300, seed=9)
    st, got, _ = gpu_forward(ctx, x, "db4", 4, S)
    assert st == 0 and "k_fwd_stream" not in ctx.last_plan(), ctx.last_plan()
    st, want = o.decompose_batch(x, get_wavelet("db4"), 4, S)
    assert_bits_equal(got, want, "row shorter than a chunk")
    # sym8 at level 7: 15 * 64 = 960 samples of context for that level alone
    x = signals(2, 20000, seed=10)
    w = get_wavelet("sym8")
    st, want = o.decompose_batch(x, w, 7, S)
    assert st == 0
    st, got, _ = gpu_forward(ctx, x, "sym8", 7, S)
    assert st == 0, ctx.last_error()
    assert "k_fwd_stream[1-5]" in ctx.last_plan() and "k_level_fwd[7-7]" in ctx.last_plan(), ctx.last_plan()
    assert_bits_equal(got, want, "mixed plan forward")
    st, back, _ = gpu_inverse(ctx, want, "sym8", S)
    assert st == 0, ctx.last_error()
    assert "k_level_inv[7-7]" in ctx.last_plan() and "k_inv_sym" in ctx.last_plan(), ctx.last_plan()
    assert_bits_equal(back, o.reconstruct_batch(want, w, S), "mixed plan inverse")


@pytest.mark.parametrize("mask_bits", [0b1000101, 0b0111111, 0b1000000, 0b0000001])
def test_symmetric_band_masks_through_tiles(ctx, mask_bits):
    """reconstructFromLevel / reconstructLevels: masked bands enter as zeros (the products stay in the sums)."""
    wn, n, j, b = "db4", 5000, 6, 3
    w = get_wavelet(wn)
    x = signals(b, n, seed=21)
    st, c = o.decompose_batch(x, w, j, S)
    assert st == 0
    z = c.copy()
    for k in range(j + 1):
        if not (mask_bits >> k) & 1:
            z[:, k, :] = 0.0
    hr, gr = _t(w.rec_lo), _t(w.rec_hi)
    out = np.full((b, n), np.nan)
    st = ctx.lib.vw_modwt_inverse_bands_batch(ctx.handle, _vp(c), b, n, j, _dp(hr), hr.size, _dp(gr), gr.size, S, mask_bits, _vp(out))
    assert st == 0, ctx.last_error()
    assert "k_inv_sym" in ctx.last_plan(), ctx.last_plan()
    assert_bits_equal(out, o.reconstruct_batch(z, w, S), "mask %s" % bin(mask_bits))


def test_symmetric_finite_check_through_tiles(ctx):
    wn, n, j, b = "db4", 6000, 5, 4
    x = signals(b, n, seed=31)
    x[2, 4321] = np.inf
    st, _, bad = gpu_forward(ctx, x, wn, j, S, N.VW_FLAG_CHECK_FINITE)
    assert st == N.VW_E_NONFINITE and bad == 2 * n + 4321
    x[2, 4321] = 0.5
    st, c = o.decompose_batch(x, get_wavelet(wn), j, S)
    c[1, 3, 17] = np.nan
    st, _, bad = gpu_inverse(ctx, c, wn, S, N.VW_FLAG_CHECK_FINITE)
    assert st == N.VW_E_NONFINITE and bad == (1 * (j + 1) + 3) * n + 17
    assert "k_inv_sym" in ctx.last_plan(), ctx.last_plan()


@pytest.mark.parametrize("thr,soft", [(-1.0, True), (0.2, False)])
def test_symmetric_swt_denoise_through_tiles(ctx, thr, soft):
    """VectorWaveSwtAdapter.denoise with SYMMETRIC: forward on tiles, threshold pass over the block, inverse on tiles."""
    w = get_wavelet("db4")
    x = signals(4, 5000, seed=41)
    st, want = o.swt_denoise_batch(x, w, 5, thr, soft, S)
    assert st == 0
    st, got, _ = gpu_denoise(ctx, x, "db4", 5, S, thr, soft)
    assert st == N.VW_OK, ctx.last_error()
    assert "k_fwd_stream" in ctx.last_plan() and "k_inv_sym" in ctx.last_plan(), ctx.last_plan()
    assert_bits_equal(got, want, "SYMMETRIC swt denoise")


def test_symmetric_many_rows_many_tiles(ctx):
    """Runs of several tiles per workgroup (the carried histories) behind a mirrored first tile."""
    w = get_wavelet("db4")
    x = signals(40, 32768, seed=51)
    st, want = o.decompose_batch(x, w, 6, S)
    assert st == 0
    st, got, _ = gpu_forward(ctx, x, "db4", 6, S)
    assert st == 0, ctx.last_error()
    assert_bits_equal(got, want, "forward 40 x 32768")
    st, back, _ = gpu_inverse(ctx, want, "db4", S)
    assert st == 0, ctx.last_error()
    assert_bits_equal(back, o.reconstruct_batch(want, w, S), "inverse 40 x 32768")


@pytest.mark.parametrize("method", ["UNIVERSAL", "MINIMAX", "BAYES"])
@pytest.mark.parametrize("soft", [True, False])
@pytest.mark.parametrize("levels", [0, 5])
def test_symmetric_wavelet_denoiser_thresholds_on_load(ctx, method, soft, levels):
    """WaveletDenoiser with SYMMETRIC (denoising/WaveletDenoiser.java:121-227): the per-level thresholds are applied as
    k_inv_sym stages the bands (levels == 0: the single-level form, MODWTTransform.inverse's pair sum)."""
    import torch
    from util import dev_wavelet_denoise
    w = get_wavelet("db4")
    x = signals(3, 5000, seed=61)
    x[1, 777] = 3.0e3  # an outlier that survives every threshold
    m = getattr(o, method)
    st, want, wthr = o.wavelet_denoise_batch(x, w, levels, m, soft, S)
    assert st == 0
    got, thr = dev_wavelet_denoise(ctx, torch.from_numpy(x).cuda(), "db4", levels, S, getattr(N, "VW_THRESHOLD_" + method), soft)
    assert "k_inv_sym" in ctx.last_plan() and "k_fwd_stream" in ctx.last_plan(), ctx.last_plan()
    assert_bits_equal(thr.cpu().numpy(), wthr, "thresholds")
    assert_bits_equal(got.cpu().numpy(), want, "SYMMETRIC WaveletDenoiser %s levels=%d" % (method, levels))


@pytest.mark.parametrize("thr,soft", [(float("nan"), True), (float("inf"), False), (0.0, True)])
def test_symmetric_swt_denoise_odd_thresholds(ctx, thr, soft):
    w = get_wavelet("sym4")
    x = signals(2, 4098, seed=71)
    st, want = o.swt_denoise_batch(x, w, 6, thr, soft, S)
    assert st == 0
    st, got, _ = gpu_denoise(ctx, x, "sym4", 6, S, thr, soft)
    assert st == N.VW_OK, ctx.last_error()
    assert "k_inv_sym" in ctx.last_plan(), ctx.last_plan()
    assert_bits_equal(got, want, "SYMMETRIC swt denoise, threshold %r" % thr)
