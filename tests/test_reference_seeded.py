"""The reference tests' own seeded inputs, replayed.

The reference's unit tests draw their random inputs from `new java.util.Random(42).nextGaussian()`
(modwt/BatchSIMDMODWTTest.java:104-151, modwt/ParallelMultiLevelMODWTTest.java:64-100,
swt/VectorWaveSwtAdapterTest.java:243-268 + :431-439).  tools/java_random.py restates that generator as the JDK
documents it (48-bit LCG, polar method with the cached second value, fdlibm's log), so these are the reference's inputs
bit for bit, without a JVM.  What the reference asserts on them are properties -- batch == per-signal and parallel ==
sequential at 1e-10, "zeroes some, keeps some", shapes -- there are no stored outputs; tests/golden/reference_seeded.npz
keeps the inputs together with the oracle's outputs on them (tools/make_golden.py seeded).

CPU tests: the generator's known values, the fixture against the generator, the reference's assertions on the oracle.
GPU tests: the HIP path on the same inputs, bit for bit against the fixture."""
import os
import sys

import numpy as np
import pytest

from oracle import oracle as o
from util import assert_bits_equal, gpu_denoise, gpu_forward, gpu_inverse
from vectorwave_amd.wavelets import get_wavelet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from java_random import JavaRandom, strict_log  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "reference_seeded.npz"))


def test_java_random_known_values():
    # new Random(42): nextDouble() and the first Gaussians every JVM prints for this seed
    assert JavaRandom(42).next_double() == 0.7275636800328681
    r = JavaRandom(42)
    assert [r.next_gaussian() for _ in range(5)] == [1.1419053154730547, 0.9194079489827879, -0.9498666368908959,
                                                      -1.1069902863993377, 0.2809776380727795]
    # new Random(0).nextGaussian()
    assert JavaRandom(0).next_gaussian() == 0.8025330637390305
    # fdlibm's log: exact where the result is representable, within one unit of the C library's elsewhere
    assert strict_log(1.0) == 0.0 and strict_log(0.5) == -0.6931471805599453
    assert abs(strict_log(2.0 ** -1060) - np.log(2.0 ** -1060)) <= np.spacing(734.0)  # the subnormal branch
    rng = np.random.default_rng(3)
    for x in rng.random(2000):
        assert abs(strict_log(x) - np.log(x)) <= np.spacing(abs(np.log(x)))


def test_fixture_inputs_are_the_generators():
    r = JavaRandom(42)
    xb = np.array([[r.next_gaussian() for _ in range(64)] for _ in range(16)])
    assert_bits_equal(GOLD["batch_db4_x"], xb, "BatchSIMDMODWTTest DB4 batch")
    r = JavaRandom(42)
    assert_bits_equal(GOLD["parallel_db4_x"], np.array([r.next_gaussian() for _ in range(128)]), "ParallelMultiLevelMODWTTest DB4")
    assert_bits_equal(GOLD["parallel_db4_x"][:64], GOLD["batch_db4_x"][0], "same seed, same stream")
    assert_bits_equal(GOLD["batch_haar7_x"], np.array([[b + t * 0.1 for t in range(32)] for b in range(7)]), "b + t * 0.1")


def test_oracle_reproduces_the_fixture_and_the_reference_assertions():
    db4, haar = get_wavelet("db4"), get_wavelet("haar")
    # BatchSIMDMODWTTest.java:104-151: the SoA batch equals the per-signal transform at 1e-10
    xb = GOLD["batch_db4_x"]
    c = o.modwt_forward_batch(xb, db4, 0)
    assert_bits_equal(c, GOLD["batch_db4_coeffs"], "forward")
    a, d = o.batch_modwt_soa(o.convert_to_soa(xb), db4, 16, 64)
    assert_bits_equal(a, GOLD["batch_db4_soa_approx"], "soa approx")
    np.testing.assert_allclose(o.convert_from_soa(a, 16, 64), c[:, 1], rtol=0, atol=1e-10)
    np.testing.assert_allclose(o.convert_from_soa(d, 16, 64), c[:, 0], rtol=0, atol=1e-10)
    # :153-183: a batch of 7 runs and its first values are not zero
    a7, _ = o.batch_modwt_soa(o.convert_to_soa(GOLD["batch_haar7_x"]), haar, 7, 32)
    back = o.convert_from_soa(a7, 7, 32)
    assert back[0, 0] != 0.0 and back[6, 0] != 0.0
    # ParallelMultiLevelMODWTTest.java:64-100: parallel == sequential at 1e-10
    st, cs = o.decompose_batch(GOLD["parallel_db4_x"][None], db4, 4, 0)
    assert st == 0
    assert_bits_equal(cs[0], GOLD["parallel_db4_coeffs"], "decompose")
    st, cp = o.parallel_decompose(GOLD["parallel_db4_x"], db4, 4, 0)
    assert st == 0
    np.testing.assert_allclose(cp, cs[0], rtol=0, atol=1e-10)
    # VectorWaveSwtAdapterTest.java:243-268: the universal soft threshold zeroes some coefficients and keeps some
    t = GOLD["swt_universal_thresholded"]
    assert (t[:4] == 0.0).any() and (t[:4] != 0.0).any()
    st, y = o.swt_denoise_batch(GOLD["swt_universal_x"][None], db4, 4, -1.0, True, 0)
    assert st == 0
    assert_bits_equal(y[0], GOLD["swt_universal_denoised"], "denoise")
    # SimplifiedAPIWorkflowTest.java:103-144: shapes and validity
    assert GOLD["workflow_coeffs"].shape == (32, 2, 777) and np.isfinite(GOLD["workflow_coeffs"]).all()
    assert GOLD["workflow_reconstructed"].shape == (32, 777)


# ---- GPU -------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_on_the_reference_seeded_inputs(ctx):
    from vectorwave_amd import _native as N
    from vectorwave_amd import BatchSIMDMODWT, BoundaryMode, ParallelMultiLevelMODWT
    # single-level batch (MODWTTransform.forwardBatch) and the SoA batch
    xb = np.ascontiguousarray(GOLD["batch_db4_x"])
    st, c, _ = gpu_forward(ctx, xb, "db4", 1, 0, N.VW_FLAG_SINGLE_LEVEL)
    assert st == 0, ctx.last_error()
    assert_bits_equal(c, GOLD["batch_db4_coeffs"], "forwardBatch on BatchSIMDMODWTTest's batch")
    soa = np.empty(16 * 64)
    BatchSIMDMODWT.convertToSoA(xb, soa, ctx)
    a, d = np.empty(16 * 64), np.empty(16 * 64)
    BatchSIMDMODWT.batchMODWTSoA(soa, a, d, "db4", 16, 64, ctx)
    assert_bits_equal(a, GOLD["batch_db4_soa_approx"], "soaApprox")
    assert_bits_equal(d, GOLD["batch_db4_soa_detail"], "soaDetail")
    x7 = np.ascontiguousarray(GOLD["batch_haar7_x"])
    soa7, a7, d7 = np.empty(7 * 32), np.empty(7 * 32), np.empty(7 * 32)
    BatchSIMDMODWT.convertToSoA(x7, soa7, ctx)
    BatchSIMDMODWT.batchMODWTSoA(soa7, a7, d7, "haar", 7, 32, ctx)
    assert_bits_equal(a7, GOLD["batch_haar7_soa_approx"], "soaApprox, batch of 7")
    assert_bits_equal(d7, GOLD["batch_haar7_soa_detail"], "soaDetail, batch of 7")
    # multi-level, sequential and "parallel"
    xp = np.ascontiguousarray(GOLD["parallel_db4_x"])
    st, c, _ = gpu_forward(ctx, xp[None], "db4", 4, 0)
    assert st == 0, ctx.last_error()
    assert_bits_equal(c[0], GOLD["parallel_db4_coeffs"], "decompose on ParallelMultiLevelMODWTTest's signal")
    r = ParallelMultiLevelMODWT(ctx=ctx).decompose(xp, "db4", BoundaryMode.PERIODIC, 4)
    assert_bits_equal(r.block(), GOLD["parallel_db4_parallel_coeffs"], "parallel decompose")
    # SWT adapter: universal soft threshold, 4 levels
    xs = np.ascontiguousarray(GOLD["swt_universal_x"])
    st, y, thr = gpu_denoise(ctx, xs[None], "db4", 4, 0, -1.0, True)
    assert st == 0, ctx.last_error()
    assert_bits_equal(thr, GOLD["swt_universal_thr"], "universal threshold")
    assert_bits_equal(y[0], GOLD["swt_universal_denoised"], "denoise on VectorWaveSwtAdapterTest's signal")
    # batch workflow: forward, soft threshold 0.1 on the details, inverse
    xw = np.ascontiguousarray(GOLD["workflow_x"])
    st, c, _ = gpu_forward(ctx, xw, "db4", 1, 0, N.VW_FLAG_SINGLE_LEVEL)
    assert st == 0, ctx.last_error()
    assert_bits_equal(c, GOLD["workflow_coeffs"], "workflow forwardBatch")
    cm = c.copy()
    for k in range(32):
        cm[k, 0] = o.soft_threshold(c[k, 0], 0.1)
    st, back, _ = gpu_inverse(ctx, cm, "db4", 0, N.VW_FLAG_SINGLE_LEVEL)
    assert st == 0, ctx.last_error()
    assert_bits_equal(back, GOLD["workflow_reconstructed"], "workflow inverseBatch")
