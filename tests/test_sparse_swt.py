"""swt/SWTResult.java sparse form: the oracle against a direct reading of SparseVector (CPU), and the GPU
compaction / expansion against the oracle, bit for bit and in index order (GPU)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as o


def _band(n, seed):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(n)
    v[rng.random(n) < 0.3] = 0.0
    return v


@pytest.mark.parametrize("n", [1, 2, 17, 256, 1000])
@pytest.mark.parametrize("thr", [0.0, 0.5, 10.0])
def test_oracle_sparse_vector(n, thr):
    d = _band(n, n)
    idx, val = o.sparse_vector(d, thr)
    keep = [i for i in range(n) if abs(d[i]) > thr]  # SWTResult.java:270-288
    assert idx.tolist() == keep and np.array_equal(val, d[keep])
    back = o.sparse_to_dense(idx, val, n)
    want = np.where(np.abs(d) > thr, d, 0.0)
    assert np.array_equal(back, want)


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,levels", [(1, 1, 1), (3, 17, 2), (2, 1023, 3), (2, 1024, 4), (3, 1025, 2), (2, 5000, 5), (1, 70001, 2)])
@pytest.mark.parametrize("thr", [0.0, 0.7, 100.0])
def test_gpu_sparsify_densify(ctx, b, n, levels, thr):
    from vectorwave_amd import _native as N
    rng = np.random.default_rng(n + levels)
    coeffs = rng.standard_normal((b, levels + 1, n))
    coeffs[rng.random(coeffs.shape) < 0.25] = 0.0
    coeffs[0, 0, 0] = -0.0
    counts = np.full(b * levels, -1, dtype=np.int64)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    assert ctx.lib.vw_swt_sparsify_batch(ctx.handle, vp(coeffs), b, n, levels, thr, vp(counts), None, None, None) == N.VW_OK
    want = [o.sparse_vector(coeffs[r, j], thr) for r in range(b) for j in range(levels)]
    assert counts.tolist() == [w[0].size for w in want]
    offsets = np.zeros(b * levels + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    total = int(offsets[-1])
    idx = np.full(max(total, 1), -1, dtype=np.int32)
    val = np.full(max(total, 1), np.nan)
    assert ctx.lib.vw_swt_sparsify_batch(ctx.handle, vp(coeffs), b, n, levels, thr, None, vp(offsets), vp(idx), vp(val)) == N.VW_OK
    for k, (wi, wv) in enumerate(want):
        lo, hi = offsets[k], offsets[k + 1]
        assert np.array_equal(idx[lo:hi], wi), "indices of band %d" % k
        assert np.array_equal(val[lo:hi].view(np.uint64), wv.view(np.uint64)), "values of band %d" % k
    out = np.full_like(coeffs, 7.0)
    assert ctx.lib.vw_swt_densify_batch(ctx.handle, vp(offsets), vp(idx), vp(val), b, n, levels, vp(out)) == N.VW_OK
    assert np.all(out[:, levels] == 7.0)  # the approximation band is the caller's
    for r in range(b):
        for j in range(levels):
            wi, wv = want[r * levels + j]
            assert np.array_equal(out[r, j].view(np.uint64), o.sparse_to_dense(wi, wv, n).view(np.uint64))


@pytest.mark.gpu
def test_swt_result_mirror(ctx):
    from vectorwave_amd import SWTResult
    rng = np.random.default_rng(5)
    n, levels = 300, 3
    a = rng.standard_normal(n)
    d = rng.standard_normal((levels, n)) * np.array([1.0, 0.1, 0.01])[:, None]
    res = SWTResult(a, d, levels, ctx)
    assert res.getLevels() == levels and res.getSignalLength() == n
    assert np.array_equal(res.getDetail(1), d[1]) and np.array_equal(res.getApproximation(), a)
    e = sum(float(v) * float(v) for v in a)
    assert res.getApproximationEnergy() == e
    total = e
    for k in range(levels):
        total += res.getDetailEnergy(k)  # SWTResult.java:168-181: approximation first, then level by level
    assert res.getTotalEnergy() == total
    sp = res.toSparse(0.05)
    assert sp.getThreshold() == 0.05
    kept = 0
    for k in range(levels):
        wi, wv = o.sparse_vector(d[k], 0.05)
        gi, gv = sp.getSparseDetail(k)
        assert np.array_equal(gi, wi) and np.array_equal(gv, wv)
        kept += wi.size
    assert sp.getCompressionRatio() == (n * (levels + 1)) / (n + 2 * kept)  # SWTResult.java:238-247
    full = sp.toFull()
    assert np.array_equal(full.getApproximation(), a)
    assert np.array_equal(full.getAllDetails(), np.where(np.abs(d) > 0.05, d, 0.0))
    assert full == SWTResult(a, np.where(np.abs(d) > 0.05, d, 0.0), levels, ctx) and full != res
    assert hash(full) == hash(SWTResult(a, np.where(np.abs(d) > 0.05, d, 0.0), levels, ctx))
    with pytest.raises(ValueError):
        SWTResult(a, d, levels + 1)
    with pytest.raises(ValueError):
        SWTResult(a, [d[0], d[1][:-1], d[2]], levels)
    with pytest.raises(ValueError):
        res.getDetail(levels)
    with pytest.raises(TypeError):
        SWTResult(None, d, levels)


@pytest.mark.gpu
def test_gpu_sparse_argument_checks(ctx):
    from vectorwave_amd import _native as N
    vp = lambda a: C.c_void_p(a.ctypes.data)
    coeffs = np.ones((1, 3, 8))
    idx = np.zeros(16, dtype=np.int32)
    val = np.zeros(16)
    bad = np.array([0, 9, 3], dtype=np.int64)        # decreasing
    assert ctx.lib.vw_swt_sparsify_batch(ctx.handle, vp(coeffs), 1, 8, 2, 0.5, None, vp(bad), vp(idx), vp(val)) == N.VW_E_LENGTH_MISMATCH
    neg = np.array([-1, 0, 3], dtype=np.int64)
    assert ctx.lib.vw_swt_densify_batch(ctx.handle, vp(neg), vp(idx), vp(val), 1, 8, 2, vp(coeffs)) == N.VW_E_NOT_POSITIVE
    # an index outside the band is dropped, not written
    offs = np.array([0, 2, 2], dtype=np.int64)
    idx[:2] = [3, 100]
    val[:2] = [5.0, 6.0]
    out = np.full((1, 3, 8), 7.0)
    assert ctx.lib.vw_swt_densify_batch(ctx.handle, vp(offs), vp(idx), vp(val), 1, 8, 2, vp(out)) == N.VW_OK
    want = np.zeros((2, 8))
    want[0, 3] = 5.0
    assert np.array_equal(out[0, :2], want) and np.all(out[0, 2] == 7.0)
    assert ctx.lib.vw_swt_sparsify_batch(ctx.handle, vp(coeffs), 1, 8, 0, 0.5, vp(offs), None, None, None) == N.VW_E_LEVEL


@pytest.mark.gpu
def test_fill_pass_with_a_lower_threshold_stays_inside_its_slots(ctx):
    """Offsets sized by a counting pass at one threshold, fill pass at a lower one: every band then holds more
    entries than its slot.  Nothing may be written at or beyond offsets[band + 1] (the last band's slot ends the
    caller's buffers), the call reports the mismatch, and the _dev form tells the real counts."""
    import torch
    from vectorwave_amd import _native as N
    vp = lambda a: C.c_void_p(a.ctypes.data)
    b, n, levels = 3, 3000, 2
    rng = np.random.default_rng(5)
    coeffs = rng.standard_normal((b, levels + 1, n))
    hi_thr, lo_thr = 1.5, 0.2
    counts = np.zeros(b * levels, dtype=np.int64)
    assert ctx.lib.vw_swt_sparsify_batch(ctx.handle, vp(coeffs), b, n, levels, hi_thr, vp(counts), None, None, None) == N.VW_OK
    offsets = np.zeros(b * levels + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    total = int(offsets[-1])
    guard = 4096
    idx = np.full(total + guard, -7, dtype=np.int32)
    val = np.full(total + guard, -7.0)
    st = ctx.lib.vw_swt_sparsify_batch(ctx.handle, vp(coeffs), b, n, levels, lo_thr, None, vp(offsets), vp(idx), vp(val))
    assert st == N.VW_E_LENGTH_MISMATCH and "entries" in ctx.last_error()
    assert np.all(idx[total:] == -7) and np.all(val[total:] == -7.0)
    # what was written is the head of each band's entry list at the lower threshold
    for k in range(b * levels):
        wi, wv = o.sparse_vector(coeffs[k // levels, k % levels], lo_thr)
        room = int(offsets[k + 1] - offsets[k])
        assert np.array_equal(idx[offsets[k]:offsets[k + 1]], wi[:room])
        assert np.array_equal(val[offsets[k]:offsets[k + 1]], wv[:room])
    # device form: guard region behind the buffers untouched, counts_dev receives the real numbers
    c_t = torch.from_numpy(coeffs).cuda()
    off_t = torch.from_numpy(offsets).cuda()
    idx_t = torch.full((total + guard,), -7, dtype=torch.int32, device="cuda")
    val_t = torch.full((total + guard,), -7.0, dtype=torch.float64, device="cuda")
    cnt_t = torch.full((b * levels,), -1, dtype=torch.int64, device="cuda")
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    st = ctx.lib.vw_swt_sparsify_batch_dev(ctx.handle, C.c_void_p(c_t.data_ptr()), b, n, levels, lo_thr, C.c_void_p(cnt_t.data_ptr()),
                                           C.c_void_p(off_t.data_ptr()), C.c_void_p(idx_t.data_ptr()), C.c_void_p(val_t.data_ptr()), sp)
    assert st == N.VW_OK
    torch.cuda.synchronize()
    assert bool((idx_t[total:] == -7).all()) and bool((val_t[total:] == -7.0).all())
    real = [o.sparse_vector(coeffs[k // levels, k % levels], lo_thr)[0].size for k in range(b * levels)]
    assert cnt_t.cpu().tolist() == real
    assert np.array_equal(idx_t[:total].cpu().numpy(), idx[:total])
