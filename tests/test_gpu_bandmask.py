"""Band-selective inverse against the oracle, bit for bit.

reconstructFromLevel / reconstructLevels (modwt/MultiLevelMODWTTransform.java:385-468) and
VectorWaveSwtAdapter.extractLevel (swt/VectorWaveSwtAdapter.java:326-348) all reconstruct from a coefficient
block in which some bands are replaced by zeros.  The library takes a band mask instead of zeroed copies; the
expected bits are the oracle's reconstruction of the block with those bands set to zero.  Every mask the three
reference methods can produce is checked, on a single-group plan (db4, short rows), on the headline tile plan
(db4 N=16384 J=6) and on the long-context streaming plan (sym8 J=8) at three run lengths."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as o
from util import P, Z, _dp, _t, _vp, assert_bits_equal, signals
from vectorwave_amd import _native as N
from vectorwave_amd.api import (BoundaryMode, MultiLevelMODWTTransform, VectorWaveSwtAdapter)
from vectorwave_amd.wavelets import get_wavelet

pytestmark = pytest.mark.gpu


def reference_masks(J):
    """(label, mask) of every partial reconstruction the reference offers for a J-level result; bit k-1 = d_k,
    bit J = a_J."""
    out = []
    for k in range(1, J + 1):  # reconstructFromLevel(k): a_J and d_k..d_J
        m = 1 << J
        for lv in range(k, J + 1):
            m |= 1 << (lv - 1)
        out.append(("fromLevel(%d)" % k, m))
    for a in range(1, J + 1):  # reconstructLevels(a, b): d_a..d_b, a_J only if b == J
        for b in range(a, J + 1):
            m = 0
            for lv in range(a, b + 1):
                m |= 1 << (lv - 1)
            if b == J:
                m |= 1 << J
            out.append(("levels(%d,%d)" % (a, b), m))
    for k in range(0, J + 1):  # extractLevel(k): one band; 0 = the approximation
        out.append(("extract(%d)" % k, (1 << J) if k == 0 else 1 << (k - 1)))
    seen, uniq = set(), []
    for label, m in out:
        if m not in seen:
            seen.add(m)
            uniq.append((label, m))
    return uniq


def oracle_masked(coeffs, w, mode, mask):
    z = coeffs.copy()
    J = coeffs.shape[1] - 1
    for k in range(J + 1):
        if not (mask >> k) & 1:
            z[:, k, :] = 0.0
    return o.reconstruct_batch(z, w, mode)


def gpu_masked(ctx, coeffs, wname, mode, mask):
    w = get_wavelet(wname)
    hr, gr = _t(w.rec_lo), _t(w.rec_hi)
    b, jp1, n = coeffs.shape
    out = np.full((b, n), np.nan)
    st = ctx.lib.vw_modwt_inverse_bands_batch(ctx.handle, _vp(coeffs), b, n, jp1 - 1, _dp(hr), hr.size, _dp(gr), gr.size, mode,
                                              int(mask), _vp(out))
    assert st == 0, ctx.last_error()
    return out


def gpu_masked_dev(ctx, c_t, wname, mode, mask):
    import torch
    w = get_wavelet(wname)
    hr, gr = _t(w.rec_lo), _t(w.rec_hi)
    b, jp1, n = c_t.shape
    out = torch.full((b, n), float("nan"), dtype=torch.float64, device=c_t.device)
    sp = C.c_void_p(torch.cuda.current_stream(c_t.device).cuda_stream)
    st = ctx.lib.vw_modwt_inverse_bands_batch_dev(ctx.handle, C.c_void_p(c_t.data_ptr()), b, n, jp1 - 1, _dp(hr), hr.size, _dp(gr),
                                                  gr.size, mode, int(mask), C.c_void_p(out.data_ptr()), sp)
    assert st == 0, ctx.last_error()
    return out.cpu().numpy()


def _check_all_masks(ctx, wn, b, n, J, mode, what):
    import torch
    w = get_wavelet(wn)
    x = signals(b, n, seed=n + J)
    st, coeffs = o.decompose_batch(x, w, J, mode)
    assert st == 0
    c_t = torch.from_numpy(coeffs).cuda()
    for label, mask in reference_masks(J):
        want = oracle_masked(coeffs, w, mode, mask)
        assert_bits_equal(gpu_masked(ctx, coeffs, wn, mode, mask), want, "%s %s host" % (what, label))
        assert_bits_equal(gpu_masked_dev(ctx, c_t, wn, mode, mask), want, "%s %s dev" % (what, label))


@pytest.mark.parametrize("mode", [P, Z])
def test_masks_short_rows_single_group(ctx, mode):
    _check_all_masks(ctx, "db4", 3, 256, 4, mode, "db4 N=256 J=4 mode=%d" % mode)


@pytest.mark.parametrize("mode", [P, Z])
def test_masks_headline_tile_plan(ctx, mode):
    _check_all_masks(ctx, "db4", 2, 16384, 6, mode, "db4 N=16384 J=6 mode=%d" % mode)


@pytest.mark.parametrize("mode", [P, Z])
@pytest.mark.parametrize("chunks", [1, 3, 1000])
def test_masks_long_context_plan(ctx, plan, chunks, mode):
    plan("sdry_chunks", chunks)
    _check_all_masks(ctx, "sym8", 2, 20000, 8, mode, "sym8 N=20000 J=8 chunks=%d mode=%d" % (chunks, mode))


@pytest.mark.parametrize("mode", [P, Z])
def test_masks_multi_group_plan(ctx, plan, mode):
    """Long-context kernels off: db20 J=8 (context 39 * 255 samples, beyond the largest halo shape) needs two halo
    groups, so the running reconstruction ping-pongs through the context's plane and the mask has to reach both."""
    plan("no_sdry", 1)
    _check_all_masks(ctx, "db20", 2, 16384, 8, mode, "db20 N=16384 J=8 multi-group mode=%d" % mode)


@pytest.mark.parametrize("mode", [BoundaryMode.PERIODIC, BoundaryMode.ZERO_PADDING])
def test_facade_partial_reconstructions_bit_exact(ctx, mode):
    """The same through the API mirror: the packing of the result containers (band order, masks built from level
    arguments) must hand the library exactly what the reference methods mean."""
    wn, n, J = "db4", 1000, 5
    w = get_wavelet(wn)
    t = MultiLevelMODWTTransform(wn, mode, ctx)
    x = signals(1, n, seed=77)
    st, coeffs = o.decompose_batch(x, w, J, int(mode))
    assert st == 0
    res = t.decompose(x[0], J)
    # decompose: d_1..d_J through getDetailCoeffsAtLevel, a_J through getApproximationCoeffs
    for lv in range(1, J + 1):
        assert_bits_equal(res.getDetailCoeffsAtLevel(lv), coeffs[0, lv - 1], "facade detail %d" % lv)
    assert_bits_equal(res.getApproximationCoeffs(), coeffs[0, J], "facade approximation")
    assert_bits_equal(t.reconstruct(res), o.reconstruct_batch(coeffs, w, int(mode))[0], "facade reconstruct")
    for k in range(1, J + 1):
        m = 1 << J
        for lv in range(k, J + 1):
            m |= 1 << (lv - 1)
        assert_bits_equal(t.reconstructFromLevel(res, k), oracle_masked(coeffs, w, int(mode), m)[0], "facade fromLevel(%d)" % k)
    for a in range(1, J + 1):
        for b in range(a, J + 1):
            m = sum(1 << (lv - 1) for lv in range(a, b + 1)) | ((1 << J) if b == J else 0)
            assert_bits_equal(t.reconstructLevels(res, a, b), oracle_masked(coeffs, w, int(mode), m)[0],
                              "facade levels(%d,%d)" % (a, b))
    ad = VectorWaveSwtAdapter(wn, mode, ctx)
    for k in range(0, J + 1):
        m = (1 << J) if k == 0 else 1 << (k - 1)
        assert_bits_equal(ad.extractLevel(x[0], J, k), oracle_masked(coeffs, w, int(mode), m)[0], "facade extractLevel(%d)" % k)
