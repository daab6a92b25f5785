"""Frozen golden vectors (tests/golden/, made by tools/make_golden.py).

CPU: the oracle must reproduce the committed vectors bit for bit (guards against compiler
or flag drift in the checker itself) and satisfy the reference's closed-form KATs.
GPU: the HIP path, through the C ABI, must reproduce the same committed vectors."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as o
from vectorwave_amd.wavelets import get_wavelet

from util import assert_bits_equal

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    z = np.load(os.path.join(GOLD, "oracle_vectors.npz"))
    out = []
    for key in z["_cases"]:
        key = str(key)
        wn, n, j, m = key.rsplit("_", 3)
        out.append((key, wn.replace("_", "."), int(n[1:]), int(j[1:]), int(m[1:])))
    return z, out


def test_reference_kats_on_oracle():
    k = json.load(open(os.path.join(GOLD, "reference_kats.json")))
    c = o.modwt_forward_batch(np.array([k["haar_level1_approx"]["signal"]]), get_wavelet("haar"), 0)
    np.testing.assert_allclose(c[0, 1], k["haar_level1_approx"]["approx"], rtol=0, atol=k["haar_level1_approx"]["tol"])
    for name, fn in (("circular_06_04", o.circular_convolve), ("zero_pad_06_04", o.zero_padding_convolve),
                     ("manual_circular", o.circular_convolve)):
        e = k[name]
        np.testing.assert_allclose(fn(e["signal"], e["filter"]), e["expected"], rtol=0, atol=e["tol"])
    e = k["level2_dilation"]
    np.testing.assert_allclose(o.circular_convolve_level(e["signal"], e["filter"], e["level"])[:2],
                               e["expected_first_two"], rtol=0, atol=e["tol"])
    e = k["time_reversed_haar"]
    np.testing.assert_allclose(o.circular_convolve(e["signal"], e["filter"])[:2], e["expected_first_two"], rtol=0,
                               atol=e["tol"])
    for n, l, want in k["max_levels"]["cases"]:
        assert o.max_levels(n, l) == want


def test_oracle_reproduces_golden_vectors():
    z, cases = _cases()
    assert len(cases) == 30
    for key, wn, n, j, mode in cases:
        w = get_wavelet(wn)
        x = z[key + "_x"]
        st, c = o.decompose_batch(x, w, j, mode)
        assert st == 0
        assert_bits_equal(c, z[key + "_coeffs"], key + " coeffs")
        assert_bits_equal(o.reconstruct_batch(c, w, mode), z[key + "_recon"], key + " recon")
        if mode != 2:
            assert_bits_equal(o.swt_denoise_batch(x, w, j, -1.0, True, mode)[1], z[key + "_denoise_soft"], key + " denoise")


@pytest.mark.gpu
def test_gpu_reproduces_golden_vectors(ctx):
    from util import gpu_denoise, gpu_forward, gpu_inverse
    z, cases = _cases()
    for key, wn, n, j, mode in cases:
        x = np.ascontiguousarray(z[key + "_x"])
        st, c, _ = gpu_forward(ctx, x, wn, j, mode)
        assert st == 0, ctx.last_error()
        assert_bits_equal(c, z[key + "_coeffs"], key + " coeffs")
        st, r, _ = gpu_inverse(ctx, np.ascontiguousarray(z[key + "_coeffs"]), wn, mode)
        assert st == 0, ctx.last_error()
        assert_bits_equal(r, z[key + "_recon"], key + " recon")
        if mode != 2:
            st, y, thr = gpu_denoise(ctx, x, wn, j, mode, -1.0, True)
            assert st == 0, ctx.last_error()
            assert_bits_equal(y, z[key + "_denoise_soft"], key + " denoise")
            assert_bits_equal(thr, z[key + "_thr"], key + " thresholds")


@pytest.mark.gpu
def test_gpu_reference_kats(ctx):
    from vectorwave_amd import BoundaryMode, MODWTTransform, MultiLevelMODWTTransform, WaveletOperations as W
    k = json.load(open(os.path.join(GOLD, "reference_kats.json")))
    e = k["haar_level1_approx"]
    r = MODWTTransform("haar", BoundaryMode.PERIODIC, ctx).forward(e["signal"])
    np.testing.assert_allclose(r.approximationCoeffs(), e["approx"], rtol=0, atol=e["tol"])
    e = k["circular_06_04"]
    np.testing.assert_allclose(W.circularConvolveMODWT(e["signal"], e["filter"], None, ctx), e["expected"], rtol=0, atol=e["tol"])
    e = k["zero_pad_06_04"]
    np.testing.assert_allclose(W.zeroPaddingConvolveMODWT(e["signal"], e["filter"], None, ctx), e["expected"], rtol=0, atol=e["tol"])
    e = k["manual_circular"]
    np.testing.assert_allclose(W.circularConvolveMODWT(e["signal"], e["filter"], None, ctx), e["expected"], rtol=0, atol=e["tol"])
    e = k["level2_dilation"]
    np.testing.assert_allclose(W.circularConvolveMODWTLevel(e["signal"], e["filter"], None, e["level"], ctx)[:2],
                               e["expected_first_two"], rtol=0, atol=e["tol"])
    e = k["time_reversed_haar"]
    np.testing.assert_allclose(W.circularConvolveMODWT(e["signal"], e["filter"], None, ctx)[:2], e["expected_first_two"], rtol=0, atol=e["tol"])
    t = MultiLevelMODWTTransform("haar", BoundaryMode.PERIODIC, ctx)
    assert t.getMaximumLevels(10000) == 9 and MultiLevelMODWTTransform.getMaxDecompositionLevels() == 10


# ---- second fixture: WaveletDenoiser and long-context level groups -------------------------------------

def _extra():
    z = np.load(os.path.join(GOLD, "denoiser_longcontext_vectors.npz"))
    dn = []
    for key in z["_denoiser_cases"]:
        key = str(key)
        wn, n, lv, m, s, mode = key.rsplit("_", 5)
        dn.append((key, wn.replace("_", "."), int(n[1:]), int(lv[1:]), int(m[1:]), int(s[1:]), int(mode[1:])))
    lc = []
    for key in z["_long_context_cases"]:
        key = str(key)
        wn, n, j, mode = key.rsplit("_", 3)
        lc.append((key, wn.replace("_", "."), int(n[1:]), int(j[1:]), int(mode[1:])))
    return z, dn, lc


def test_oracle_reproduces_extra_golden_vectors():
    z, dn, lc = _extra()
    assert len(dn) == 8 and len(lc) == 4
    for key, wn, n, levels, method, soft, mode in dn:
        st, y, thr = o.wavelet_denoise_batch(z[key + "_x"], get_wavelet(wn), levels, method, bool(soft), mode,
                                             fixed_threshold=0.2)
        assert st == 0
        assert_bits_equal(thr, z[key + "_thr"], key + " thresholds")
        assert_bits_equal(y, z[key + "_y"], key + " denoised")
    for key, wn, n, j, mode in lc:
        w = get_wavelet(wn)
        st, c = o.decompose_batch(z[key + "_x"], w, j, mode)
        assert st == 0
        assert_bits_equal(c, z[key + "_coeffs"], key + " coeffs")
        assert_bits_equal(o.reconstruct_batch(c, w, mode), z[key + "_recon"], key + " recon")


@pytest.mark.gpu
def test_gpu_reproduces_extra_golden_vectors(ctx):
    from util import gpu_forward, gpu_inverse
    from vectorwave_amd import BoundaryMode, ThresholdType, WaveletDenoiser
    z, dn, lc = _extra()
    for key, wn, n, levels, method, soft, mode in dn:
        den = WaveletDenoiser(wn, BoundaryMode(mode), ctx)
        y, thr = den.denoise_block(np.ascontiguousarray(z[key + "_x"]), levels, method,
                                   ThresholdType.SOFT if soft else ThresholdType.HARD, 0.2, return_thresholds=True)
        assert_bits_equal(thr, z[key + "_thr"], key + " thresholds")
        assert_bits_equal(y, z[key + "_y"], key + " denoised")
    for key, wn, n, j, mode in lc:
        st, c, _ = gpu_forward(ctx, np.ascontiguousarray(z[key + "_x"]), wn, j, mode)
        assert st == 0, ctx.last_error()
        assert_bits_equal(c, z[key + "_coeffs"], key + " coeffs")
        st, r, _ = gpu_inverse(ctx, np.ascontiguousarray(z[key + "_coeffs"]), wn, mode)
        assert st == 0, ctx.last_error()
        assert_bits_equal(r, z[key + "_recon"], key + " recon")


# ---- third fixture: streaming denoiser / streaming transforms / sparse SWT result ---------------------

def _streaming():
    z = np.load(os.path.join(GOLD, "streaming_sparse_vectors.npz"))
    cases = []
    for key in z["_streaming_cases"]:
        key = str(key)
        wn, n, ring, est, method, mult, soft, mode = key.split("_")
        cases.append((key, wn, int(n[1:]), int(ring[1:]), int(est[1:]), int(method[1:]), float(mult[1:]), int(soft[1:]), int(mode[1:])))
    return z, cases


def test_oracle_reproduces_streaming_golden_vectors():
    z, cases = _streaming()
    assert len(cases) == 6
    for key, wn, n, ring, est, method, mult, soft, mode in cases:
        w = get_wavelet(wn)
        state = o.StreamingDenoiserState(2, ring)
        for k, blk in enumerate(z[key + "_x"]):
            st, y, _ = o.streaming_denoise_batch(state, blk, w, method, bool(soft), mult, est, mode)
            assert st == 0
            assert_bits_equal(y, z[key + "_y"][k], key + " block %d" % k)
            assert_bits_equal(state.level, z[key + "_level"][k], key + " level %d" % k)
            assert_bits_equal(state.window, z[key + "_ring"][k], key + " ring %d" % k)
    w = get_wavelet("db4")
    t = o.StreamingTransformState(w, 128, 0)
    assert_bits_equal(np.concatenate([t.process(z["xform_x"]), t.flush()]), z["xform_blocks"], "streaming transform blocks")
    assert_bits_equal(o.stream_multilevel(z["xform_x"], w, 128, 3, 0)[1], z["xform_multilevel_blocks"], "multi-level blocks")
    idx, val = o.sparse_vector(z["sparse_dense"], 0.6)
    assert np.array_equal(idx, z["sparse_idx"])
    assert_bits_equal(val, z["sparse_val"], "sparse values")


@pytest.mark.gpu
def test_gpu_reproduces_streaming_golden_vectors(ctx):
    from vectorwave_amd import (BoundaryMode, MODWTStreamingDenoiser, MODWTStreamingTransform, NoiseEstimation, SWTResult,
                                ThresholdMethod, ThresholdType)
    z, cases = _streaming()
    for key, wn, n, ring, est, method, mult, soft, mode in cases:
        den = (MODWTStreamingDenoiser.Builder().wavelet(wn).boundaryMode(BoundaryMode(mode)).noiseWindowSize(ring)
               .thresholdMethod(ThresholdMethod(method)).thresholdType(ThresholdType.SOFT if soft else ThresholdType.HARD)
               .thresholdMultiplier(mult).noiseEstimation(NoiseEstimation(est)).streams(2).context(ctx).build())
        with den:
            for k, blk in enumerate(z[key + "_x"]):
                assert_bits_equal(den.denoise(np.ascontiguousarray(blk)), z[key + "_y"][k], key + " block %d" % k)
                assert_bits_equal(den.getEstimatedNoiseLevel(), z[key + "_level"][k], key + " level %d" % k)
                assert_bits_equal(den.getNoiseWindow()[0], z[key + "_ring"][k], key + " ring %d" % k)
    got = []
    with MODWTStreamingTransform.create("db4", BoundaryMode.PERIODIC, 128, ctx=ctx) as t:
        t.subscribe(lambda r: got.append([r.detailCoeffs(), r.approximationCoeffs()]))
        t.process(z["xform_x"][:300])
        t.process(z["xform_x"][300:])
    assert_bits_equal(np.array(got), z["xform_blocks"], "streaming transform blocks")
    got = []
    with MODWTStreamingTransform.createMultiLevel("db4", BoundaryMode.PERIODIC, 128, 3, streams=2, ctx=ctx) as t:
        t.subscribe(got.append)
        t.process(np.stack([z["xform_x"], z["xform_x"]]))
    assert_bits_equal(np.array(got)[:, 1], z["xform_multilevel_blocks"], "multi-level blocks")
    d = z["sparse_dense"]
    sp = SWTResult(np.zeros(d.size), [d], 1, ctx).toSparse(0.6)
    idx, val = sp.getSparseDetail(0)
    assert np.array_equal(idx, z["sparse_idx"])
    assert_bits_equal(val, z["sparse_val"], "sparse values")
