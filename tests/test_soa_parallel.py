"""modwt/BatchSIMDMODWT.java (structure-of-arrays batch) and modwt/ParallelMultiLevelMODWT.java.

CPU: the oracle against what the reference's own tests assert (BatchSIMDMODWTTest.java,
ParallelMultiLevelMODWTTest.java).  GPU: the HIP path against the oracle, bit for bit -- including the
Haar path's literal 0.5 constants, which differ from MODWTTransform in the last bits."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as o
from vectorwave_amd.wavelets import get_wavelet

from util import assert_bits_equal, signals

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pattern(b, n):
    # BatchSIMDMODWTTest.java:61-66
    return np.array([[(bb + 1) * np.sin(2 * np.pi * t / n) for t in range(n)] for bb in range(b)])


def test_soa_layout_kat():
    k = json.load(open(os.path.join(GOLD, "reference_kats.json")))["soa_layout"]  # BatchSIMDMODWTTest.java:28-52
    rows = np.array(k["rows"])
    soa = o.convert_to_soa(rows)
    assert soa.tolist() == k["soa"]
    assert np.array_equal(o.convert_from_soa(soa, 3, 4), rows)


@pytest.mark.parametrize("wn,b,n", [("haar", 8, 16), ("db4", 8, 32), ("db2", 8, 32), ("haar", 7, 16), ("db4", 5, 32)])
def test_soa_matches_sequential_transform(wn, b, n):
    # BatchSIMDMODWTTest.java:55-102 (Haar), :104-151 (DB4), :153-183 (batch not a multiple of the vector length)
    w = get_wavelet(wn)
    x = _pattern(b, n)
    a, d = o.batch_modwt_soa(o.convert_to_soa(x), w, b, n)
    c = o.modwt_forward_batch(x, w, 0)
    np.testing.assert_allclose(o.convert_from_soa(a, b, n), c[:, 1], rtol=0, atol=1e-10)
    np.testing.assert_allclose(o.convert_from_soa(d, b, n), c[:, 0], rtol=0, atol=1e-10)


@pytest.mark.parametrize("wn,mode,levels", [("haar", 0, 4), ("db4", 0, 3), ("db4", 1, 3), ("db4", 2, 2)])
def test_parallel_matches_sequential(wn, mode, levels):
    # ParallelMultiLevelMODWTTest.java:31-133: same coefficients as MultiLevelMODWTTransform (1e-10); any
    # non-PERIODIC mode runs the zero-padding convolution (ParallelMultiLevelMODWT.java:100-131)
    w = get_wavelet(wn)
    x = signals(1, 256, seed=levels)[0]
    st, c = o.parallel_decompose(x, w, levels, mode)
    assert st == 0
    st2, want = o.decompose_batch(x[None], w, levels, 0 if mode == 0 else 1)
    assert st2 == 0
    np.testing.assert_allclose(c, want[0], rtol=0, atol=1e-10)


def test_parallel_level_limits():
    # ParallelMultiLevelMODWTTest.java:203-225 and ParallelMultiLevelMODWT.java:218-248
    x = np.array([1.0, 2.0, 3.0, 4.0])
    assert o.parallel_decompose(x, get_wavelet("haar"), 1)[0] == 0
    assert o.parallel_decompose(x, get_wavelet("haar"), 0)[0] != 0
    assert o.parallel_decompose(x, get_wavelet("haar"), 10)[0] != 0
    assert o.parallel_max_levels(8, 8) == 0 and o.parallel_max_levels(2, 2) == 0 and o.parallel_max_levels(3, 2) == 2


# ---- GPU -------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("wn,b,n", [("haar", 8, 16), ("haar", 7, 33), ("db2", 8, 32), ("db2", 3, 5), ("db4", 8, 32), ("db4", 5, 1000),
                                    ("sym8", 130, 257), ("coif1", 4, 64), ("db4", 300, 7)])
def test_gpu_soa_batch_bit_exact(ctx, wn, b, n):
    from vectorwave_amd import BatchSIMDMODWT
    w = get_wavelet(wn)
    x = signals(b, n, seed=b + n)
    soa = np.empty(b * n)
    BatchSIMDMODWT.convertToSoA(x, soa, ctx)
    assert_bits_equal(soa, o.convert_to_soa(x), "convertToSoA")
    a, d = np.empty(b * n), np.empty(b * n)
    BatchSIMDMODWT.batchMODWTSoA(soa, a, d, w, b, n, ctx)
    wa, wd = o.batch_modwt_soa(soa, w, b, n)
    assert_bits_equal(a, wa, "soaApprox %s" % wn)
    assert_bits_equal(d, wd, "soaDetail %s" % wn)
    back = np.empty((b, n))
    BatchSIMDMODWT.convertFromSoA(a, back, ctx)
    assert_bits_equal(back, o.convert_from_soa(wa, b, n), "convertFromSoA")


@pytest.mark.gpu
@pytest.mark.parametrize("run_len", [0, 8, 16, 32, -1])
@pytest.mark.parametrize("wn,b,n", [("haar", 1024, 100), ("haar", 515, 33), ("db2", 1030, 70), ("db4", 2048, 129), ("db4", 777, 40),
                                    ("coif1", 600, 50), ("db6", 514, 64), ("sym8", 1000, 90), ("db10", 520, 77), ("sym9", 300, 60), ("sym5", 258, 41), ("sym7", 512, 30), ("sym11", 300, 50),
                                    ("db4", 6, 5000)])
def test_gpu_soa_run_kernel_variants(ctx, plan, wn, b, n, run_len):
    """The register-window kernel (8, 16 or 32 samples per lane; pairs of signals per lane when the batch is even, single
    signals when it is odd) and the one-output-per-lane kernel (run_len -1) give the oracle's bits: several blocks along
    the batch, several along time, the first block wrapping around the row end (modwt/BatchSIMDMODWT.java:86-274)."""
    from vectorwave_amd import BatchSIMDMODWT
    if run_len < 0:
        plan("no_soa_runs", 1)
    elif run_len > 0:
        plan("soa_run_len", run_len)
    w = get_wavelet(wn)
    x = signals(b, n, seed=b + n)
    soa = o.convert_to_soa(x)
    a, d = np.empty(b * n), np.empty(b * n)
    BatchSIMDMODWT.batchMODWTSoA(soa, a, d, w, b, n, ctx)
    wa, wd = o.batch_modwt_soa(soa, w, b, n)
    assert_bits_equal(a, wa, "soaApprox %s %d x %d" % (wn, b, n))
    assert_bits_equal(d, wd, "soaDetail %s %d x %d" % (wn, b, n))


@pytest.mark.gpu
def test_gpu_soa_unequal_filter_lengths_rejected(ctx):
    # generalBatchMODWTSoA loops l < lowPass.length over highPass too (BatchSIMDMODWT.java:240-249): with
    # bior4.4 (9 / 7 taps) the reference indexes out of bounds; the library reports it instead
    from vectorwave_amd import BatchSIMDMODWT
    soa = np.zeros(4 * 64)
    with pytest.raises(IndexError):
        BatchSIMDMODWT.batchMODWTSoA(soa, np.empty(256), np.empty(256), "bior4.4", 4, 64, ctx)


@pytest.mark.gpu
def test_gpu_parallel_multilevel(ctx):
    from vectorwave_amd import BoundaryMode, InvalidArgumentException, InvalidSignalException, ParallelMultiLevelMODWT
    t = ParallelMultiLevelMODWT(ctx=ctx)
    for wn, mode, levels, n in (("haar", 0, 4, 256), ("db4", 0, 3, 256), ("db4", 1, 3, 300), ("db4", 2, 2, 300), ("sym8", 0, 5, 4000)):
        x = signals(1, n, seed=n + levels)[0]
        st, want = o.parallel_decompose(x, get_wavelet(wn), levels, mode)
        assert st == 0
        r = t.decompose(x, wn, BoundaryMode(mode), levels)
        assert r.getLevels() == levels
        assert_bits_equal(r.block(), want, "parallel decompose %s mode %d" % (wn, mode))
    x = np.array([1.0, 2.0, 3.0, 4.0])
    assert t.decompose(x, "haar", BoundaryMode.PERIODIC, 1).getLevels() == 1
    with pytest.raises(InvalidSignalException):
        t.decompose(np.array([]), "haar", BoundaryMode.PERIODIC, 1)
    with pytest.raises(InvalidArgumentException):
        t.decompose(x, "haar", BoundaryMode.PERIODIC, 0)
    with pytest.raises(InvalidArgumentException):
        t.decompose(x, "haar", BoundaryMode.PERIODIC, 10)
    with pytest.raises(InvalidSignalException):
        t.decompose(np.array([1.0, np.inf, 2.0, 3.0]), "haar", BoundaryMode.PERIODIC, 1)
