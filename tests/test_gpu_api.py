"""The host mirror of the reference API on the GPU: the reference's own unit tests for
MODWTTransform / MultiLevelMODWTTransform / VectorWaveSwtAdapter restated against
vectorwave_amd (reference test file:line in each comment; paths under
src/test/java/ai/prophetizo/wavelet/)."""
import math

import numpy as np
import pytest

from vectorwave_amd import (BoundaryMode, InvalidArgumentException, InvalidSignalException, MODWTResult,
                            MODWTTransform, MultiLevelMODWTTransform, VectorWaveSwtAdapter, get_wavelet)

from util import signals

pytestmark = pytest.mark.gpu
PER, ZP, SYM = BoundaryMode.PERIODIC, BoundaryMode.ZERO_PADDING, BoundaryMode.SYMMETRIC


def test_constructor_validation(ctx):
    # modwt/MODWTTransformTest.java:29-42
    with pytest.raises(TypeError):
        MODWTTransform(None, PER, ctx)
    with pytest.raises(TypeError):
        MODWTTransform("haar", None, ctx)
    with pytest.raises(InvalidArgumentException) as e:
        MODWTTransform("haar", 5, ctx)
    assert e.value.getErrorCode() == "CFG_003"
    for m in (PER, ZP, SYM):
        MODWTTransform("haar", m, ctx)


@pytest.mark.parametrize("n", [8, 5, 1])
def test_single_level_perfect_reconstruction(ctx, n):
    # modwt/MODWTTransformTest.java:70-128
    t = MODWTTransform("haar", PER, ctx)
    x = signals(1, n, seed=n)[0]
    r = t.forward(x)
    assert r.getSignalLength() == n and r.isValid()
    np.testing.assert_allclose(t.inverse(r), x, rtol=0, atol=1e-12)


def test_invalid_signals(ctx):
    # modwt/MODWTTransformTest.java:131-154
    t = MODWTTransform("haar", PER, ctx)
    with pytest.raises(TypeError):
        t.forward(None)
    with pytest.raises(InvalidSignalException) as e:
        t.forward([])
    assert e.value.getErrorCode() == "VAL_006"
    for bad in (np.nan, np.inf, -np.inf):
        with pytest.raises(InvalidSignalException) as e:
            t.forward([1.0, bad, 3.0])
        assert e.value.getErrorCode() == "VAL_003"
    with pytest.raises(TypeError):
        t.inverse(None)


def test_symmetric_and_zero_padding_bounded(ctx):
    # modwt/MODWTTransformTest.java:85-100; MultiLevelMODWTTransformTest.java:74-88
    x = np.arange(1.0, 9.0)
    t = MODWTTransform("haar", SYM, ctx)
    assert np.abs(t.inverse(t.forward(x)) - x).max() < 1.1 * 8
    m = MultiLevelMODWTTransform("haar", SYM, ctx)
    assert np.abs(m.reconstruct(m.decompose(x, 2)) - x).max() < 3.1 * 8
    z = MODWTTransform("haar", ZP, ctx)
    r = z.inverse(z.forward(x))
    np.testing.assert_allclose(r[1:-1], x[1:-1], atol=1e-10)


@pytest.mark.parametrize("wn,n,j", [("haar", 8, 3), ("haar", 7, 2), ("db4", 32, 3), ("haar", 16, None),
                                    ("haar", 100, None), ("haar", 1000, None), ("haar", 4, None), ("db4", 1000, 3)])
def test_multilevel_reconstruction(ctx, wn, n, j):
    # modwt/MultiLevelMODWTTransformTest.java:43-71, :169-200, :235-251; swt/SWTOptimizationTest.java:66-80
    t = MultiLevelMODWTTransform(wn, PER, ctx)
    x = signals(1, n, seed=n)[0]
    res = t.decompose(x) if j is None else t.decompose(x, j)
    assert res.getSignalLength() == n
    assert res.getLevels() == (t.getMaximumLevels(n) if j is None else j)
    np.testing.assert_allclose(t.reconstruct(res), x, rtol=0, atol=1e-10)


def test_multilevel_level_validation(ctx):
    # modwt/MultiLevelMODWTTransformTest.java:203-218, :254-291
    t = MultiLevelMODWTTransform("haar", PER, ctx)
    x = signals(1, 64)[0]
    for bad in (0, 20, -1):
        with pytest.raises(InvalidArgumentException) as e:
            t.decompose(x, bad)
        assert e.value.getErrorCode() == "CFG_004"
    assert MultiLevelMODWTTransform.getMaxDecompositionLevels() == 10
    assert t.getMaximumLevels(10000) == 9
    # finiteness is validated before the level count (modwt/MultiLevelMODWTTransform.java:216-244)
    xb = x.copy()
    xb[3] = np.nan
    with pytest.raises(InvalidSignalException):
        t.decompose(xb, 20)
    with pytest.raises(InvalidSignalException):
        t.decompose([], 1)


def test_partial_reconstructions(ctx):
    # modwt/MultiLevelMODWTTransform.java:385-468 semantics
    t = MultiLevelMODWTTransform("db4", PER, ctx)
    x = signals(1, 256, seed=3)[0]
    res = t.decompose(x, 4)
    full = t.reconstruct(res)
    np.testing.assert_allclose(t.reconstructFromLevel(res, 1), full, atol=1e-12)
    bands = sum(t.reconstructLevels(res, k, k) for k in range(1, 4)) + t.reconstructLevels(res, 4, 4)
    np.testing.assert_allclose(bands, x, atol=1e-8)
    with pytest.raises(InvalidArgumentException):
        t.reconstructLevels(res, 0, 2)
    with pytest.raises(InvalidArgumentException):
        t.reconstructFromLevel(res, 5)


def test_batch_equals_per_signal(ctx):
    # modwt/BatchSIMDMODWTTest.java:55-182: batch == per-signal forward at 1e-10 (here: exact);
    # integration/SimplifiedAPIWorkflowTest.java:103-144: 32 x 777 DB4 forwardBatch/inverseBatch
    for wn, b, n in (("haar", 8, 16), ("db4", 16, 64), ("db4", 7, 64), ("db4", 32, 777), ("db4", 8, 128)):
        t = MODWTTransform(wn, PER, ctx)
        xs = signals(b, n, seed=b + n)
        batch = t.forwardBatch(list(xs))
        assert len(batch) == b
        for i in range(b):
            one = t.forward(xs[i])
            assert np.array_equal(batch[i].approximationCoeffs(), one.approximationCoeffs())
            assert np.array_equal(batch[i].detailCoeffs(), one.detailCoeffs())
        back = t.inverseBatch(batch)
        for i in range(b):
            np.testing.assert_allclose(back[i], xs[i], atol=1e-10)
    # ragged batch falls back to per-signal calls
    t = MODWTTransform("db4", PER, ctx)
    rag = t.forwardBatch([np.ones(10), np.ones(20), np.ones(30)])
    assert [r.getSignalLength() for r in rag] == [10, 20, 30]
    assert t.forwardBatch([]) == []
    with pytest.raises(TypeError):
        t.forwardBatch(None)
    # forwardBatchOptimized skips input validation; the NaN surfaces when results are built
    xs = signals(4, 64)
    xs[2, 7] = np.nan
    with pytest.raises(InvalidSignalException):
        t.forwardBatch(list(xs))


def test_swt_adapter(ctx):
    # swt/VectorWaveSwtAdapterTest.java:23-42 (DB4 J=3 N=64 PR 1e-10)
    a = VectorWaveSwtAdapter("db4", ctx=ctx)
    assert a.getBoundaryMode() == PER
    x = signals(1, 64, seed=64)[0]
    res = a.forward(x, 3)
    np.testing.assert_allclose(a.inverse(res), x, atol=1e-10)
    # :79-104 hard threshold: |c| <= thr -> 0, others unchanged
    before = res.getDetailCoeffsAtLevel(1)
    a.applyThreshold(res, 1, 0.05, False)
    after = res.getDetailCoeffsAtLevel(1)
    assert np.all(after[np.abs(before) <= 0.05] == 0.0)
    assert np.array_equal(after[np.abs(before) > 0.05], before[np.abs(before) > 0.05])
    # soft on the approximation (level 0)
    ab = res.getApproximationCoeffs()
    a.applyThreshold(res, 0, 0.1, True)
    aa = res.getApproximationCoeffs()
    np.testing.assert_array_equal(aa, np.where(np.abs(ab) > 0.1, np.sign(ab) * (np.abs(ab) - 0.1), 0.0))


def test_swt_universal_threshold_and_denoise(ctx):
    # swt/VectorWaveSwtAdapterTest.java:243-268: DB4 J=4 N=256, noise sigma=0.1: zeroes some, keeps some
    a = VectorWaveSwtAdapter("db4", PER, ctx)
    n = 256
    t = np.arange(n)
    rng = np.random.default_rng(42)
    x = np.sin(2 * np.pi * t / 32.0) + 0.1 * rng.standard_normal(n)
    res = a.forward(x, 4)
    d1 = res.getDetailCoeffsAtLevel(1)
    thr = a.universalThreshold(res)
    srt = np.sort(np.abs(d1))
    assert thr == ((srt[n // 2 - 1] + srt[n // 2]) / 2.0 / 0.6745) * math.sqrt(2 * math.log(n))
    a.applyUniversalThreshold(res, True)
    nz = sum(int((res.getDetailCoeffsAtLevel(k) == 0).sum()) for k in range(1, 5))
    kept = sum(int((res.getDetailCoeffsAtLevel(k) != 0).sum()) for k in range(1, 5))
    assert nz > 0 and kept > 0
    # :336-356 denoise(-1) == denoise(-999); step-by-step == one-call
    y1 = a.denoise(x, 4, -1, True)
    y2 = a.denoise(x, 4, -999, True)
    assert np.array_equal(y1, y2)
    assert np.array_equal(a.inverse(res), y1)
    # :315-333 PERIODIC round trip RMSE < 1e-8, other modes < 1.0
    # (the reference's signal generator, :423-430, Haar J=2 on 64 samples)
    i64 = np.arange(64)
    s64 = np.sin(2 * np.pi * i64 / 16.0) + 0.5 * np.cos(4 * np.pi * i64 / 16.0)
    for mode, tol in ((PER, 1e-8), (ZP, 1.0), (SYM, 1.0)):
        am = VectorWaveSwtAdapter("haar", mode, ctx)
        r = am.inverse(am.forward(s64, 2))
        assert math.sqrt(np.mean((r - s64) ** 2)) < tol
    # :205-217 sum of extractLevel(0..3) == signal at 1e-8
    total = sum(a.extractLevel(x, 3, k) for k in range(4))
    np.testing.assert_allclose(total, x, atol=1e-8)
    # batch denoise == per-signal denoise
    xs = signals(5, 480, seed=9)
    yb = a.denoiseBatch(xs, 4)
    for i in range(5):
        assert np.array_equal(yb[i], a.denoise(xs[i], 4))


def test_result_containers(ctx):
    # swt/VectorWaveSwtAdapterTest.java:399-419: NaN -> isValid() == false
    t = MultiLevelMODWTTransform("haar", PER, ctx)
    res = t.decomposeMutable(signals(1, 32)[0], 2)
    assert res.isValid()
    res.getMutableDetailCoeffs(1)[3] = np.nan
    assert not res.isValid()
    res.getMutableDetailCoeffs(1)[3] = 0.0
    # getters clone, getMutable* alias
    d = res.getDetailCoeffsAtLevel(1)
    d[0] = 123.0
    assert res.getDetailCoeffsAtLevel(1)[0] != 123.0
    with pytest.raises(ValueError):
        res.getDetailCoeffsAtLevel(3)
    e = res.getRelativeEnergyDistribution()
    assert abs(e.sum() - 1.0) < 1e-12 and e.size == 3
    imm = res.toImmutable()
    ei = imm.getRelativeEnergyDistribution()
    assert abs(ei[0] - e[2]) < 1e-15  # immutable: [approx, d1..dJ]; mutable: [d1..dJ, approx]
    with pytest.raises(InvalidSignalException):
        MODWTResult.create([1.0, np.nan], [0.0, 0.0])
    with pytest.raises(ValueError):
        MODWTResult.create([1.0], [0.0, 0.0])


@pytest.mark.gpu
def test_page_locked_host_buffers(ctx):
    """vw_host_alloc / vw_host_free: the host-pointer entry points give the same bits from page-locked buffers"""
    import ctypes as C
    from oracle import oracle as o
    from vectorwave_amd import _native as N
    from vectorwave_amd.wavelets import get_wavelet
    from util import assert_bits_equal, gpu_forward, signals
    x = signals(5, 1000, seed=3)
    xp = ctx.host_array((5, 1000))
    xp[:] = x
    st, got, _ = gpu_forward(ctx, xp, "db4", 3, 0)
    st2, want = o.decompose_batch(x, get_wavelet("db4"), 3, 0)
    assert st == 0 and st2 == 0
    assert_bits_equal(got, want, "forward from a page-locked buffer")
    assert ctx.host_array((0, 4)).size == 0
    ptr = C.c_void_p()
    assert ctx.lib.vw_host_alloc(ctx.handle, 4096, C.byref(ptr)) == N.VW_OK and ptr.value
    assert ctx.lib.vw_host_free(ctx.handle, ptr) == N.VW_OK
    assert ctx.lib.vw_host_free(ctx.handle, None) == N.VW_OK
    assert ctx.lib.vw_host_alloc(ctx.handle, 16, None) == N.VW_E_NULL


@pytest.mark.parametrize("mode", [PER, ZP, SYM])
@pytest.mark.parametrize("wn,n,j", [("haar", 64, 3), ("db4", 256, 4), ("db4", 777, 3), ("sym8", 3000, 5), ("bior2.2", 200, 2)])
def test_facade_packing_bit_exact(ctx, wn, n, j, mode):
    """What the mirror classes hand out -- which array is the detail, which the approximation, which level a band
    belongs to, what a partial reconstruction zeroes -- compared with the oracle bit for bit, so a swapped or shifted
    band in the host-side packing cannot hide behind a reconstruction tolerance
    (modwt/MODWTTransform.java:123-291, modwt/MultiLevelMODWTTransform.java:200-468, swt/VectorWaveSwtAdapter.java:122-348)."""
    from oracle import oracle as o
    from util import assert_bits_equal
    w = get_wavelet(wn)
    xs = signals(3, n, seed=n + j)
    x = xs[0]
    # single level
    t = MODWTTransform(wn, mode, ctx)
    want1 = o.modwt_forward_batch(xs, w, int(mode))
    r = t.forward(x)
    assert_bits_equal(r.detailCoeffs(), want1[0, 0], "forward: detail")
    assert_bits_equal(r.approximationCoeffs(), want1[0, 1], "forward: approximation")
    assert_bits_equal(t.inverse(r), o.modwt_inverse_batch(want1[:1], w, int(mode), batch_variant=0)[0], "inverse")
    if len(w.rec_lo) == len(w.rec_hi):
        for i, rb in enumerate(t.forwardBatch(xs)):
            assert_bits_equal(rb.detailCoeffs(), want1[i, 0], "forwardBatch: detail %d" % i)
            assert_bits_equal(rb.approximationCoeffs(), want1[i, 1], "forwardBatch: approximation %d" % i)
    # multi-level
    m = MultiLevelMODWTTransform(wn, mode, ctx)
    st, want = o.decompose_batch(xs, w, j, int(mode))
    assert st == 0
    for res in (m.decompose(x, j), m.decomposeMutable(x, j)):
        for lev in range(1, j + 1):
            assert_bits_equal(res.getDetailCoeffsAtLevel(lev), want[0, lev - 1], "decompose: detail level %d" % lev)
        assert_bits_equal(res.getApproximationCoeffs(), want[0, j], "decompose: approximation")
    res = m.decompose(x, j)
    if len(w.rec_lo) == len(w.rec_hi):
        assert_bits_equal(m.reconstruct(res), o.reconstruct_batch(want[:1], w, int(mode))[0], "reconstruct")
        for k in range(1, j + 1):
            z = want[:1].copy()
            z[0, :k - 1] = 0.0  # reconstructFromLevel(k): details finer than k dropped
            assert_bits_equal(m.reconstructFromLevel(res, k), o.reconstruct_batch(z, w, int(mode))[0], "reconstructFromLevel(%d)" % k)
            z = np.zeros_like(want[:1])
            z[0, k - 1] = want[0, k - 1]  # reconstructLevels(k, k): that detail band alone ...
            if k == j:
                z[0, j] = want[0, j]  # ... with the approximation when the range reaches the top level
            assert_bits_equal(m.reconstructLevels(res, k, k), o.reconstruct_batch(z, w, int(mode))[0], "reconstructLevels(%d,%d)" % (k, k))
        for i, rb in enumerate(m.decomposeBatch(xs, j)):
            for lev in range(1, j + 1):
                assert_bits_equal(rb.getDetailCoeffsAtLevel(lev), want[i, lev - 1], "decomposeBatch: row %d level %d" % (i, lev))
            assert_bits_equal(rb.getApproximationCoeffs(), want[i, j], "decomposeBatch: row %d approximation" % i)
    # SWT adapter
    if mode != SYM and len(w.rec_lo) == len(w.rec_hi):
        a = VectorWaveSwtAdapter(wn, mode, ctx)
        sres = a.forward(x, j)
        for lev in range(1, j + 1):
            assert_bits_equal(sres.getDetailCoeffsAtLevel(lev), want[0, lev - 1], "swt forward: level %d" % lev)
        for soft in (True, False):
            st, wd = o.swt_denoise_batch(xs, w, j, -1.0, soft, int(mode))
            assert st == 0
            assert_bits_equal(a.denoise(x, j, -1, soft), wd[0], "swt denoise soft=%s" % soft)
        for k in range(0, j + 1):
            z = np.zeros_like(want[:1])
            band = j if k == 0 else k - 1  # extractLevel(0) keeps the approximation alone
            z[0, band] = want[0, band]
            assert_bits_equal(a.extractLevel(x, j, k), o.reconstruct_batch(z, w, int(mode))[0], "extractLevel(%d)" % k)
