"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Bar: bit-for-bit (the kernels round every product and
sum separately and add taps in the reference's order)."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import oracle as o
from vectorwave_amd import _native as N
from vectorwave_amd.wavelets import get_wavelet

from util import P, S, Z, assert_bits_equal, gpu_denoise, gpu_forward, gpu_inverse, signals

pytestmark = pytest.mark.gpu

MODES = [P, Z, S]

# (wavelet, N, J or None for max, B)
ML_CASES = [
    ("haar", 8, 3, 2), ("haar", 7, 2, 3), ("haar", 64, 5, 2), ("db2", 64, 3, 3), ("db4", 64, 3, 2),
    ("db4", 777, None, 3), ("db4", 1000, 3, 5), ("db4", 1024, 6, 4), ("db4", 4096, 6, 3), ("db4", 4100, 6, 2),
    ("db4", 5001, 5, 2), ("db4", 16384, 6, 2), ("db10", 1000, 4, 2), ("sym8", 4096, 8, 2), ("coif3", 777, 5, 2),
    ("db10", 9999, 8, 2), ("sym8", 12288, 8, 1), ("bior4.4", 500, 4, 2), ("bior1.3", 300, 3, 2), ("bior2.2", 257, 4, 2),
    ("rbio1.3", 40, 4, 2), ("rbio2.4", 100, 5, 2), ("dmey", 2000, 5, 1), ("coif17", 4000, 5, 1), ("db20", 3000, 6, 1),
]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("wn,n,j,b", ML_CASES)
def test_multilevel_forward_inverse_bit_exact(ctx, wn, n, j, b, mode):
    w = get_wavelet(wn)
    if j is None:
        j = o.max_levels(n, len(w.dec_lo))
    x = signals(b, n, seed=n + j)
    st, want = o.decompose_batch(x, w, j, mode)
    assert st == 0
    st, got, _ = gpu_forward(ctx, x, wn, j, mode)
    assert st == 0, ctx.last_error()
    assert_bits_equal(got, want, "forward %s N=%d J=%d mode=%d" % (wn, n, j, mode))
    if len(w.rec_hi) < len(w.rec_lo) and mode != S:
        st, _, _ = gpu_inverse(ctx, want, wn, mode)
        assert st == N.VW_E_LENGTH_MISMATCH  # the reference indexes out of bounds here
        return
    wantx = o.reconstruct_batch(want, w, mode)
    st, gotx, _ = gpu_inverse(ctx, want, wn, mode)
    assert st == 0, ctx.last_error()
    assert_bits_equal(gotx, wantx, "inverse %s N=%d J=%d mode=%d" % (wn, n, j, mode))
    if mode == P and w.orthogonal and wn != "dmey":
        # reconstruction vs the reference's reconstruction is exact; vs the input it is
        # bounded by the tap tables' own orthogonality error
        tol = 1e-10 if wn in ("haar", "db2", "db4") else 1e-3
        assert np.abs(gotx - x).max() < tol * max(1.0, np.abs(x).max())


@pytest.mark.parametrize("mode", [P, Z])
@pytest.mark.parametrize("wgs", [1, 6, 24])
@pytest.mark.parametrize("wn,n,j,b,tile", [("db4", 16384, 6, 3, None), ("db4", 16390, 6, 2, "2,1"), ("haar", 9000, 8, 2, "1,1"),
                                           ("sym4", 20000, 5, 2, None), ("bior2.2", 12288, 6, 2, "2,1"),
                                           ("db2", 40000, 7, 1, "4,1"), ("db4", 5121, 5, 3, "1,1")])
def test_streaming_runs_bit_exact(ctx, plan, wn, n, j, b, tile, wgs, mode):
    """The tile-streaming kernels: one workgroup walks several consecutive tiles of a row and carries
    the level histories in LDS.  The plan option stream_wgs (wanted workgroups per launch) is lowered so that small
    batches exercise whole-row runs (1), runs that start mid-row (6) and short runs (24)."""
    plan("stream_wgs", wgs)
    if tile:
        ppt, hpt = (int(v) for v in tile.split(","))
        plan("fwd_tile", ppt * 100 + hpt)
        plan("inv_tile", ppt * 100 + hpt)
    w = get_wavelet(wn)
    x = signals(b, n, seed=n + j + wgs)
    st, want = o.decompose_batch(x, w, j, mode)
    assert st == 0
    st, got, _ = gpu_forward(ctx, x, wn, j, mode)
    assert st == 0, ctx.last_error()
    assert_bits_equal(got, want, "streaming forward %s N=%d J=%d wgs=%d" % (wn, n, j, wgs))
    wantx = o.reconstruct_batch(want, w, mode)
    st, gotx, _ = gpu_inverse(ctx, want, wn, mode)
    assert st == 0, ctx.last_error()
    assert_bits_equal(gotx, wantx, "streaming inverse %s N=%d J=%d wgs=%d" % (wn, n, j, wgs))
    # denoise rides the same kernels (threshold applied as W_j is staged, heads saved thresholded)
    st, wantd = o.swt_denoise_batch(x, w, j, -1.0, True, mode)
    assert st == 0
    st, gotd, _ = gpu_denoise(ctx, x, wn, j, mode, -1.0, True)
    assert st == 0, ctx.last_error()
    assert_bits_equal(gotd, wantd, "streaming denoise")
    # first non-finite index is reported across tile boundaries of a run
    xb = x.copy()
    bad_at = (b - 1) * n + (n * 2) // 3
    xb.reshape(-1)[bad_at] = np.nan
    st, _, bad = gpu_forward(ctx, xb, wn, j, mode, N.VW_FLAG_CHECK_FINITE)
    assert st == N.VW_E_NONFINITE and bad == bad_at


@pytest.mark.parametrize("mode", [P, Z])
@pytest.mark.parametrize("chunks", [1, 3, 1000])
@pytest.mark.parametrize("wn,n,j,b", [("sym8", 20000, 8, 2), ("db10", 30001, 8, 1), ("coif3", 12288, 7, 2), ("db20", 9000, 6, 2),
                                      ("sym8", 3000, 7, 3), ("bior4.4", 10000, 8, 2)])
def test_long_context_streaming_bit_exact(ctx, plan, wn, n, j, b, chunks, mode):
    """Level groups whose context exceeds one chunk run the long-context streaming kernel: runs of tiles
    warmed up by tiles computed with the stores off.  The plan option sdry_chunks sets the run length: 1 (every tile
    warms up on its own), 3 (runs start mid-row), 1000 (whole rows; the warm-up wraps around the row)."""
    plan("sdry_chunks", chunks)
    w = get_wavelet(wn)
    x = signals(b, n, seed=n + j + chunks)
    st, want = o.decompose_batch(x, w, j, mode)
    assert st == 0
    st, got, _ = gpu_forward(ctx, x, wn, j, mode)
    assert st == 0, ctx.last_error()
    assert_bits_equal(got, want, "long-context forward %s N=%d J=%d chunks=%d" % (wn, n, j, chunks))
    if len(w.rec_hi) >= len(w.rec_lo):
        wantx = o.reconstruct_batch(want, w, mode)
        st, gotx, _ = gpu_inverse(ctx, want, wn, mode)
        assert st == 0, ctx.last_error()
        assert_bits_equal(gotx, wantx, "long-context inverse %s N=%d J=%d chunks=%d" % (wn, n, j, chunks))
        st, wantd = o.swt_denoise_batch(x, w, j, -1.0, False, mode)
        assert st == 0
        st, gotd, _ = gpu_denoise(ctx, x, wn, j, mode, -1.0, False)
        assert st == 0, ctx.last_error()
        assert_bits_equal(gotd, wantd, "long-context denoise")
        cb = want.copy()
        bad_c = cb.size - n // 3
        cb.reshape(-1)[bad_c] = np.nan
        st, _, bad = gpu_inverse(ctx, cb, wn, mode, N.VW_FLAG_CHECK_FINITE)
        assert st == N.VW_E_NONFINITE and bad == bad_c
    xb = x.copy()
    bad_at = (b - 1) * n + n // 2 + 1
    xb.reshape(-1)[bad_at] = np.inf
    st, _, bad = gpu_forward(ctx, xb, wn, j, mode, N.VW_FLAG_CHECK_FINITE)
    assert st == N.VW_E_NONFINITE and bad == bad_at


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("wn,n,j,b", [("db4", 16384, 6, 3), ("sym8", 4096, 5, 2), ("haar", 1000, 4, 4), ("bior4.4", 2000, 4, 2),
                                      ("db10", 9999, 8, 2), ("coif17", 4000, 5, 1)])
def test_fma_mode_within_north_star_tolerance(ctx, wn, n, j, b, mode):
    """VW_FLAG_FMA trades bit parity for fused multiply-adds.  Tolerance (the benchmark's stated one):
    |gpu - ref| <= 1e-12 * max(|ref|, max|x|) per coefficient and per reconstructed sample."""
    w = get_wavelet(wn)
    x = signals(b, n, seed=3 * n + j)
    st, want = o.decompose_batch(x, w, j, mode)
    assert st == 0
    scale = np.abs(x).max()
    st, got, _ = gpu_forward(ctx, x, wn, j, mode, N.VW_FLAG_FMA)
    assert st == 0, ctx.last_error()
    assert (np.abs(got - want) <= 1e-12 * np.maximum(np.abs(want), scale)).all()
    wantx = o.reconstruct_batch(want, w, mode)
    st, gotx, _ = gpu_inverse(ctx, want, wn, mode, N.VW_FLAG_FMA)
    assert st == 0, ctx.last_error()
    assert (np.abs(gotx - wantx) <= 1e-12 * np.maximum(np.abs(wantx), scale)).all()
    if mode == S:
        # SYMMETRIC runs the generic kernels, which have no fused mode: still bit-exact
        assert_bits_equal(got, want, "symmetric forward ignores the flag")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("wn", ["haar", "db4", "sym8", "bior4.4"])
@pytest.mark.parametrize("n,b", [(1, 2), (5, 3), (8, 1), (63, 4), (64, 4), (1000, 32), (2049, 2)])
def test_single_level_bit_exact(ctx, wn, n, b, mode):
    # BASELINE config 1 is the (1000, 32) DB4 PERIODIC case
    w = get_wavelet(wn)
    x = signals(b, n, seed=n)
    want = o.modwt_forward_batch(x, w, mode)
    st, got, _ = gpu_forward(ctx, x, wn, 1, mode, N.VW_FLAG_SINGLE_LEVEL)
    assert st == 0, ctx.last_error()
    assert_bits_equal(got, want, "single forward")
    for variant, flag in ((0, 0), (1, N.VW_FLAG_SYMMETRIC_BATCH_VARIANT)):
        wantx = o.modwt_inverse_batch(want, w, mode, batch_variant=variant)
        st, gotx, _ = gpu_inverse(ctx, want, wn, mode, N.VW_FLAG_SINGLE_LEVEL | flag)
        assert st == 0, ctx.last_error()
        assert_bits_equal(gotx, wantx, "single inverse variant %d" % variant)


@pytest.mark.parametrize("mode", [P, Z])
@pytest.mark.parametrize("soft", [True, False])
@pytest.mark.parametrize("wn,n,j,b,thr", [("db4", 480, 4, 64, -1.0), ("haar", 480, 4, 16, -1.0), ("db4", 256, 4, 8, -999.0),
                                          ("bior4.4", 480, 4, 5, -1.0), ("db2", 31, 3, 7, -1.0), ("sym8", 1365, 4, 3, -1.0),
                                          ("db4", 481, 3, 8, -1.0), ("sym8", 4097, 5, 3, -1.0), ("db4", 9000, 6, 2, -1.0),
                                          ("db4", 480, 4, 8, 0.25), ("coif3", 1000, 3, 4, 0.0)])
def test_swt_denoise_bit_exact(ctx, wn, n, j, b, thr, soft, mode):
    # BASELINE config 5 shape: J=4 universal soft threshold on 480-sample blocks
    w = get_wavelet(wn)
    t = np.arange(n)
    rng = np.random.default_rng(n + j)
    x = np.ascontiguousarray(np.sin(2 * np.pi * t / 32.0)[None] + 0.5 * rng.standard_normal((b, n)))
    st, want = o.swt_denoise_batch(x, w, j, thr, soft, mode)
    assert st == 0
    st, got, thrs = gpu_denoise(ctx, x, wn, j, mode, thr, soft)
    assert st == 0, ctx.last_error()
    if thr < 0:
        st, c = o.decompose_batch(x, w, j, mode)
        want_thr = np.array([o.universal_threshold(c[i, 0]) for i in range(b)])
        assert_bits_equal(thrs, want_thr, "universal thresholds")
    assert_bits_equal(got, want, "denoise")


def test_median_with_duplicates_and_zeros(ctx):
    # even N with duplicated middle elements, all-zero band, odd N
    lib = ctx.lib
    for n, fill in ((10, None), (10, 0.0), (11, None), (4096, None), (4098, None), (70000, None)):
        rng = np.random.default_rng(n)
        d1 = np.round(rng.standard_normal((3, n)), 1) if fill is None else np.full((3, n), fill)
        coeffs = np.zeros((3, 2, n))
        coeffs[:, 0] = d1
        thr = np.empty(3)
        s2 = math.sqrt(2 * math.log(n))
        st = lib.vw_swt_universal_threshold_batch(ctx.handle, C.c_void_p(coeffs.ctypes.data), 3, n, 1, s2,
                                                  C.c_void_p(thr.ctypes.data))
        assert st == 0, ctx.last_error()
        want = np.array([o.universal_threshold(d1[i]) for i in range(3)])
        assert_bits_equal(thr, want, "median n=%d" % n)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 63, 64, 65, 127, 128, 256, 257, 480, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4095,
                               4097, 5000, 8192, 8193, 16384, 16385, 18431, 18432, 18433, 32767, 32768, 32769,
                               (2049, 0), (4096, 0), (5000, 0), (16384, 0), (18433, 0), (5000, 4096), (2049, 2048)])
def test_median_selection_edge_cases(ctx, plan, n):
    # every row-length class of the selection kernels (registers of one wave up to 2048 samples, of one workgroup
    # up to 4096 / 8192 / 16384 / 32768, radix select reading global memory beyond; a pair (n, cap) sends the row
    # through the 256-lane radix select instead, staged in LDS up to 18432 samples -- or up to cap, if not zero --
    # and reading global memory beyond), with the value patterns that stress an order-statistic search: few
    # distinct values, one outlier, denormals next to huge values, powers of two (keys that differ in one bit),
    # signed zeros, values that differ only in the last mantissa bit
    if isinstance(n, tuple):
        n, cap = n
        plan("no_wg_select", 1)
        if cap:
            plan("select_lds_cap", cap)
    rng = np.random.default_rng(n)
    rows = [
        rng.standard_normal(n),
        np.round(rng.standard_normal(n), 0),
        np.full(n, -3.5),
        np.concatenate([np.zeros(n - 1), [7.0]]),
        rng.choice([5e-324, 1e-310, 1e-300, 1.0, 1e300, 1.7e308], n),
        2.0 ** rng.integers(-40, 40, n),
        rng.choice([0.0, -0.0, 1.0, -1.0], n),
        1.0 + rng.integers(0, 3, n) * 2.0 ** -52,
        np.arange(n, dtype=np.float64) - n // 2,
        np.where(np.arange(n) % 2 == 0, 1.0, 2.0),
    ]
    d1 = np.ascontiguousarray(np.array(rows))
    b = d1.shape[0]
    coeffs = np.zeros((b, 2, n))
    coeffs[:, 0] = d1
    thr = np.empty(b)
    s2 = math.sqrt(2 * math.log(n)) if n > 1 else 0.0
    st = ctx.lib.vw_swt_universal_threshold_batch(ctx.handle, C.c_void_p(coeffs.ctypes.data), b, n, 1, s2, C.c_void_p(thr.ctypes.data))
    assert st == 0, ctx.last_error()
    want = np.array([o.estimate_noise_sigma(d1[i]) * s2 for i in range(b)])
    assert_bits_equal(thr, want, "median n=%d" % n)


def test_thresholds_and_raw_convolutions(ctx):
    from vectorwave_amd import WaveletOperations as W
    c = signals(1, 1001, seed=1)[0]
    assert_bits_equal(W.softThreshold(c, 0.3, ctx), o.soft_threshold(c, 0.3), "soft")
    assert_bits_equal(W.hardThreshold(c, 0.3, ctx), o.hard_threshold(c, 0.3), "hard")
    with pytest.raises(ValueError):
        W.softThreshold(c, -1.0, ctx)
    # internal/ScalarOpsTest.java:225-274, :245-258
    np.testing.assert_allclose(W.circularConvolveMODWT([1, 2, 3, 4], [0.6, 0.4], None, ctx), [2.2, 1.6, 2.6, 3.6], atol=1e-12)
    np.testing.assert_allclose(W.zeroPaddingConvolveMODWT([1, 2, 3, 4], [0.6, 0.4], None, ctx), [0.6, 1.6, 2.6, 3.6], atol=1e-12)
    out = W.circularConvolveMODWTLevel(np.arange(1.0, 9.0), [0.7, 0.3], None, 2, ctx)
    assert abs(out[0] - (1 * 0.7 + 7 * 0.3)) < 1e-10 and abs(out[1] - (2 * 0.7 + 8 * 0.3)) < 1e-10
    x = signals(1, 300, seed=4)[0]
    f = np.asarray(get_wavelet("db4").dec_lo)
    assert_bits_equal(W.circularConvolveMODWT(x, f, None, ctx), o.circular_convolve(x, f), "circ")
    assert_bits_equal(W.zeroPaddingConvolveMODWT(x, f, None, ctx), o.zero_padding_convolve(x, f), "zero")
    assert_bits_equal(W.symmetricConvolveMODWT(x, f, None, ctx), o.symmetric_convolve(x, f), "sym")
    assert_bits_equal(W.circularConvolveMODWTLevel(x, f, None, 3, ctx), o.circular_convolve_level(x, f, 3), "level")
    # sparse filter (internal/VectorOpsTest.java:453-464)
    assert_bits_equal(W.circularConvolveMODWT(x, [0.5, 0.0, 0.5, 0.0], None, ctx),
                      o.circular_convolve(x, [0.5, 0.0, 0.5, 0.0]), "sparse")


def test_status_codes(ctx):
    x = signals(2, 64)
    # levels 0 / 20 rejected (modwt/MultiLevelMODWTTransformTest.java:203-218)
    assert gpu_forward(ctx, x, "haar", 0, P)[0] == N.VW_E_LEVEL
    assert gpu_forward(ctx, x, "haar", 20, P)[0] == N.VW_E_LEVEL
    assert gpu_forward(ctx, x, "db4", 4, P)[0] == 0  # (8-1)*8+1 = 57 <= 64
    assert gpu_forward(ctx, x, "db4", 5, P)[0] == N.VW_E_LEVEL  # (8-1)*16+1 = 113 > 64
    assert gpu_forward(ctx, x, "haar", 2, 7)[0] == N.VW_E_BOUNDARY
    # NaN detection with the first bad flat index (util/ValidationUtils.java:106-117)
    xb = x.copy()
    xb[1, 5] = np.nan
    xb[1, 9] = np.inf
    st, _, bad = gpu_forward(ctx, xb, "haar", 2, P, N.VW_FLAG_CHECK_FINITE)
    assert st == N.VW_E_NONFINITE and bad == 64 + 5
    st, _, bad = gpu_forward(ctx, x, "haar", 2, P, N.VW_FLAG_CHECK_FINITE)
    assert st == 0 and bad == -1
    st, c = o.decompose_batch(x, get_wavelet("haar"), 2, P)
    c[0, 1, 3] = np.inf
    st, _, bad = gpu_inverse(ctx, c, "haar", P, N.VW_FLAG_CHECK_FINITE)
    assert st == N.VW_E_NONFINITE and bad == 64 + 3
    # empty batch / empty signal
    st, _, _ = gpu_forward(ctx, np.zeros((0, 64)), "haar", 2, P)
    assert st == 0
    st, _, _ = gpu_forward(ctx, np.zeros((2, 0)), "haar", 1, P, N.VW_FLAG_SINGLE_LEVEL)
    assert st == N.VW_E_EMPTY


@pytest.mark.parametrize("mode", [P, Z])
@pytest.mark.parametrize("variant", ["runs", "runs1", "runs2", "runs3", "runs5", "runs16"])
@pytest.mark.parametrize("wn,n,j,b", [("db4", 16384, 6, 3), ("db4", 2560, 5, 2), ("db4", 6002, 6, 2), ("haar", 8192, 8, 2),
                                      ("haar", 20480, 7, 1), ("db2", 12290, 7, 2), ("sym4", 4096, 4, 3), ("db4", 2048, 1, 2),
                                      ("db4", 30000, 2, 1), ("db4", 65536, 6, 1)])
def test_forward_tile_kernel_families_bit_exact(ctx, plan, wn, n, j, b, variant, mode):
    """The streaming forward over runs of the planner's length and of at most 1, 2, 3, 5 and 16 tiles must reproduce the
    oracle: coefficients of every band, with and without the finite check, first tiles (wrap / zeros left of the row),
    carried histories, rows that are not a whole number of tiles and tile counts that do not divide into equal runs (13
    tiles in runs of at most 5: 5 + 4 + 4)."""
    # a small batch would otherwise get one-tile runs whatever the cap: the planner wants thousands of workgroups
    plan("stream_wgs", 1)
    if len(variant) > 4:
        plan("fwd_chunks", int(variant[4:]))
    w = get_wavelet(wn)
    x = signals(b, n, seed=n + 3 * j)
    st, want = o.decompose_batch(x, w, j, mode)
    assert st == 0
    st, got, _ = gpu_forward(ctx, x, wn, j, mode)
    assert st == 0, ctx.last_error()
    assert_bits_equal(got, want, "%s forward %s N=%d J=%d" % (variant, wn, n, j))
    assert "k_fwd_stream" in ctx.last_plan()
    st, got, bad = gpu_forward(ctx, x, wn, j, mode, N.VW_FLAG_CHECK_FINITE)
    assert st == 0 and bad == -1
    assert_bits_equal(got, want, "%s checked forward" % variant)
    xb = x.copy()
    bad_at = (b - 1) * n + (n * 2) // 3
    xb.reshape(-1)[bad_at] = np.nan
    st, _, bad = gpu_forward(ctx, xb, wn, j, mode, N.VW_FLAG_CHECK_FINITE)
    assert st == N.VW_E_NONFINITE and bad == bad_at


@pytest.mark.parametrize("mode", [P, Z])
@pytest.mark.parametrize("variant", ["ring", "carry2", "carry3", "carry8", "blk", "fused"])
@pytest.mark.parametrize("wn,n,j,b", [("db4", 16384, 6, 3), ("db4", 2560, 5, 2), ("db4", 6002, 6, 2), ("haar", 8192, 8, 2),
                                      ("haar", 20480, 7, 1), ("db2", 12290, 7, 2), ("sym4", 4096, 4, 3), ("db4", 2048, 1, 2),
                                      ("db4", 30000, 2, 1), ("bior1.3", 9000, 6, 2)])
def test_inverse_tile_kernel_families_bit_exact(ctx, plan, wn, n, j, b, variant, mode):
    """The inverse tile kernels behind one plan -- the LDS-DMA ring (default on 16-byte-aligned rows of 2560+ samples), the
    ring over runs of 2, 3 and 8 tiles with carried right contexts (opt-in), register-blocked with register landing, and
    the unblocked k_inv_fused -- must each reproduce the oracle: plain
    inverse, inverse with the finite check, thresholding inverse (denoise) and band-masked inverse, on interior
    tiles, on the last tile of a row (wrap / zeros) and on rows that are not a whole number of tiles."""
    import ctypes as C
    if variant == "blk":
        plan("no_ring", 1)
    elif variant == "fused":
        plan("no_blk", 1)
    elif variant.startswith("carry"):
        plan("carry_chunks", int(variant[5:]))
    w = get_wavelet(wn)
    x = signals(b, n, seed=n + j)
    st, coeffs = o.decompose_batch(x, w, j, mode)
    assert st == 0
    want = o.reconstruct_batch(coeffs, w, mode)
    st, got, _ = gpu_inverse(ctx, coeffs, wn, mode)
    assert st == 0, ctx.last_error()
    assert_bits_equal(got, want, "%s inverse %s N=%d J=%d" % (variant, wn, n, j))
    st, got, bad = gpu_inverse(ctx, coeffs, wn, mode, N.VW_FLAG_CHECK_FINITE)
    assert st == 0 and bad == -1
    assert_bits_equal(got, want, "%s checked inverse" % variant)
    cb = coeffs.copy()
    bad_at = ((b - 1) * (j + 1) + j // 2) * n + (n * 3) // 4
    cb.reshape(-1)[bad_at] = np.inf
    st, _, bad = gpu_inverse(ctx, cb, wn, mode, N.VW_FLAG_CHECK_FINITE)
    assert st == N.VW_E_NONFINITE and bad == bad_at
    st, wantd = o.swt_denoise_batch(x, w, j, -1.0, True, mode)
    assert st == 0
    st, gotd, _ = gpu_denoise(ctx, x, wn, j, mode, -1.0, True)
    assert st == 0, ctx.last_error()
    assert_bits_equal(gotd, wantd, "%s denoise" % variant)
    # a band mask: keep the approximation and every second detail band
    mask = (1 << j) | sum(1 << k for k in range(0, j, 2))
    z = coeffs.copy()
    for k in range(j):
        if not (mask >> k) & 1:
            z[:, k, :] = 0.0
    hr, gr = np.ascontiguousarray(w.rec_lo, dtype=np.float64), np.ascontiguousarray(w.rec_hi, dtype=np.float64)
    out = np.full((b, n), np.nan)
    dp = C.POINTER(C.c_double)
    st = ctx.lib.vw_modwt_inverse_bands_batch(ctx.handle, C.c_void_p(coeffs.ctypes.data), b, n, j, hr.ctypes.data_as(dp), hr.size,
                                              gr.ctypes.data_as(dp), gr.size, mode, mask, C.c_void_p(out.ctypes.data))
    assert st == 0, ctx.last_error()
    assert_bits_equal(out, o.reconstruct_batch(z, w, mode), "%s masked inverse" % variant)
