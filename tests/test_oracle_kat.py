"""Pins the CPU oracle (oracle/vw_oracle.c) against the known-answer vectors and
properties held by the reference's own unit tests.  The reference keeps no
numeric golden files: its closed-form KATs are replayed here literally, with
the reference test file:line each one comes from (paths under
src/test/java/ai/prophetizo/wavelet/).  CPU only."""
import math

import numpy as np
import pytest

from oracle import oracle as o
from vectorwave_amd.wavelets import get_wavelet

from util import P, S, Z, signals

H = 0.7071067811865475  # the value TimeReversedFilterTest uses for h


def test_haar_level1_approx_kat():
    # modwt/MODWTPercivalWaldenValidationTest.java:73-82
    c = o.modwt_forward_batch(np.array([[1.0, 2.0, 3.0, 4.0]]), get_wavelet("haar"), P)
    np.testing.assert_allclose(c[0, 1], [2.5, 1.5, 2.5, 3.5], rtol=0, atol=1e-12)


def test_forward_matches_hand_formula():
    # modwt/MODWTPercivalWaldenValidationTest.java:91-134: W_t = sum_l g_l X_{(t-l+N)%N}
    x = np.array([2.0, 4.0, -1.0, 3.0, 0.5, -2.5, 1.5, 6.0])
    for wn in ("haar", "db2", "db4"):
        w = get_wavelet(wn)
        c = o.modwt_forward_batch(x[None], w, P)[0]
        n = x.size
        for t in range(n):
            a = sum((w.dec_lo[l] / math.sqrt(2)) * x[(t - l + 8 * n) % n] for l in range(len(w.dec_lo)))
            d = sum((w.dec_hi[l] / math.sqrt(2)) * x[(t - l + 8 * n) % n] for l in range(len(w.dec_hi)))
            assert abs(c[1, t] - a) < 1e-12 and abs(c[0, t] - d) < 1e-12


def test_scalarops_circular_and_zero_pad_kat():
    # internal/ScalarOpsTest.java:225-274
    np.testing.assert_allclose(o.circular_convolve([1, 2, 3, 4], [0.6, 0.4]),
                               [1 * 0.6 + 4 * 0.4, 2 * 0.6 + 1 * 0.4, 3 * 0.6 + 2 * 0.4, 4 * 0.6 + 3 * 0.4], atol=1e-10)
    np.testing.assert_allclose(o.zero_padding_convolve([1, 2, 3, 4], [0.6, 0.4]),
                               [0.6, 2 * 0.6 + 0.4, 3 * 0.6 + 2 * 0.4, 4 * 0.6 + 3 * 0.4], atol=1e-10)


def test_scalarops_level2_dilation_kat():
    # internal/ScalarOpsTest.java:245-258: out[0] = 1*0.7 + 7*0.3, out[1] = 2*0.7 + 8*0.3
    out = o.circular_convolve_level(np.arange(1.0, 9.0), [0.7, 0.3], 2)
    assert abs(out[0] - (1 * 0.7 + 7 * 0.3)) < 1e-10
    assert abs(out[1] - (2 * 0.7 + 8 * 0.3)) < 1e-10


def test_level_overflow_guard():
    # internal/ScalarOpsTest.java:365-373: level >= 32 rejected
    with pytest.raises(ValueError):
        o.circular_convolve_level(np.arange(8.0), [0.5, 0.5], 32)


def test_vectorops_sparse_filter_zero_skip():
    # internal/VectorOpsTest.java:453-464: filter {0.5, 0, 0.5, 0}; scalar == vector order
    x = signals(1, 256, seed=3)[0]
    f = [0.5, 0.0, 0.5, 0.0]
    a = o.circular_convolve_scalar(x, f)
    b = o.circular_convolve_vector_order(x, f)
    assert np.array_equal(a, b)
    assert abs(a[5] - (0.5 * x[5] + 0.5 * x[3])) < 1e-15


def test_time_reversed_filter_kat():
    # modwt/TimeReversedFilterTest.java:20-50: out[0] = h*1 + h*8, out[1] = h*2 + h*1
    x = np.arange(1.0, 9.0)
    for fn in (o.circular_convolve_scalar, o.circular_convolve_vector_order):
        out = fn(x, [H, H])
        assert abs(out[0] - (H * 1 + H * 8)) < 1e-10
        assert abs(out[1] - (H * 2 + H * 1)) < 1e-10


def test_circular_shift_equivariance():
    # modwt/TimeReversedFilterTest.java:53-88
    x = signals(1, 64, seed=5)
    w = get_wavelet("db4")
    c0 = o.modwt_forward_batch(x, w, P)
    c1 = o.modwt_forward_batch(np.roll(x, 3, axis=1), w, P)
    np.testing.assert_allclose(np.roll(c0, 3, axis=2), c1, atol=1e-12)


def test_manual_circular_convolution():
    # modwt/MODWTMathematicalValidationTest.java:296-316: filter {0.5, -0.5}
    x = np.array([1.0, 4.0, 2.0, 8.0])
    out = o.circular_convolve(x, [0.5, -0.5])
    np.testing.assert_allclose(out, [0.5 * 1 - 0.5 * 8, 0.5 * 4 - 0.5 * 1, 0.5 * 2 - 0.5 * 4, 0.5 * 8 - 0.5 * 2])


@pytest.mark.parametrize("wn,n", [("haar", 8), ("haar", 5), ("haar", 1), ("db2", 8), ("db4", 8)])
def test_single_level_perfect_reconstruction(wn, n):
    # modwt/MODWTTransformTest.java:70-128, MODWTPercivalWaldenValidationTest.java:200-228 (1e-12 / 1e-10)
    x = signals(1, n, seed=n)
    w = get_wavelet(wn)
    c = o.modwt_forward_batch(x, w, P)
    r = o.modwt_inverse_batch(c, w, P, batch_variant=0)
    np.testing.assert_allclose(r, x, atol=1e-10)


def test_energy_conservation_and_linearity():
    # modwt/MODWTPercivalWaldenValidationTest.java:172-195, :314-345
    w = get_wavelet("haar")
    x = signals(2, 64, seed=9)
    c = o.modwt_forward_batch(x, w, P)
    np.testing.assert_allclose((c ** 2).sum(axis=(1, 2)), (x ** 2).sum(axis=1), rtol=1e-10)
    a, b = 2.5, -1.25
    cl = o.modwt_forward_batch((a * x[0] + b * x[1])[None], w, P)[0]
    np.testing.assert_allclose(cl, a * c[0] + b * c[1], atol=1e-12)


def test_zero_padding_interior_preserved():
    # modwt/MODWTPercivalWaldenValidationTest.java:367-389
    w = get_wavelet("haar")
    x = signals(1, 32, seed=2)
    c = o.modwt_forward_batch(x, w, Z)
    r = o.modwt_inverse_batch(c, w, Z, batch_variant=0)
    np.testing.assert_allclose(r[0, 1:-1], x[0, 1:-1], atol=1e-10)
    assert 0 < np.abs(r[0] - x[0]).max() < 2.0


@pytest.mark.parametrize("wn,n,j", [("haar", 8, 3), ("haar", 7, 2), ("db4", 32, 3), ("haar", 16, None),
                                    ("haar", 100, None), ("haar", 1000, None), ("haar", 4, None), ("db4", 1000, 3)])
def test_multilevel_perfect_reconstruction(wn, n, j):
    # modwt/MultiLevelMODWTTransformTest.java:43-71, :169-200, :235-251;
    # MultiLevelMODWTFilterTruncationTest.java:69-88; swt/SWTOptimizationTest.java:66-80 (1e-10)
    w = get_wavelet(wn)
    if j is None:
        j = o.max_levels(n, len(w.dec_lo))
    x = signals(1, n, seed=n)
    st, c = o.decompose_batch(x, w, j, P)
    assert st == 0
    r = o.reconstruct_batch(c, w, P)
    np.testing.assert_allclose(r, x, atol=1e-10)


def test_max_levels():
    # modwt/MultiLevelMODWTTransformTest.java:254-291: getMaximumLevels(10000) == 9 for Haar
    assert o.max_levels(10000, 2) == 9
    assert o.max_levels(2, 2) == 0
    assert o.max_levels(8, 8) == 0
    assert o.max_levels(9, 8) == 1
    assert o.max_levels(480, 8) == 7


def test_levels_rejected():
    # modwt/MultiLevelMODWTTransformTest.java:203-218: levels 0 / 20 rejected
    w = get_wavelet("haar")
    x = signals(1, 64)
    assert o.decompose_batch(x, w, 0, P)[0] == 104
    assert o.decompose_batch(x, w, 20, P)[0] == 104


def test_symmetric_bounded_error():
    # modwt/MODWTTransformTest.java:85-100 (1.1), MultiLevelMODWTTransformTest.java:74-88 (3.1)
    w = get_wavelet("haar")
    x = np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0])[None]
    c = o.modwt_forward_batch(x, w, S)
    r = o.modwt_inverse_batch(c, w, S, batch_variant=0)
    assert np.abs(r - x).max() < 1.1 * np.abs(x).max()
    st, cm = o.decompose_batch(x, w, 2, S)
    rm = o.reconstruct_batch(cm, w, S)
    assert np.abs(rm - x).max() < 3.1 * np.abs(x).max()


def test_threshold_semantics():
    # swt/VectorWaveSwtAdapterTest.java:79-104 (hard: |c| <= thr -> 0); internal/VectorOpsTest.java:470+
    c = np.array([-3.0, -1.0, -0.5, 0.0, 0.5, 1.0, 3.0])
    np.testing.assert_array_equal(o.hard_threshold(c, 1.0), [-3.0, 0, 0, 0, 0, 0, 3.0])
    np.testing.assert_array_equal(o.soft_threshold(c, 1.0), [-2.0, 0, 0, 0, 0, 0, 2.0])
    np.testing.assert_array_equal(o.apply_threshold(c, 1.0, True), [-2.0, 0, 0, 0, 0, 0, 2.0])
    np.testing.assert_array_equal(o.apply_threshold(c, 1.0, False), [-3.0, 0, 0, 0, 0, 0, 3.0])


def test_universal_threshold_and_denoise():
    # swt/VectorWaveSwtAdapterTest.java:243-268 (DB4 J=4 N=256 zeroes some, keeps some),
    # :336-356 (denoise(-1) == denoise(-999)), :23-42 (DB4 J=3 N=64 PR 1e-10)
    w = get_wavelet("db4")
    n = 256
    t = np.arange(n)
    rng = np.random.default_rng(42)
    x = (np.sin(2 * np.pi * t / 32.0) + 0.1 * rng.standard_normal(n))[None]
    st, c = o.decompose_batch(x, w, 4, P)
    thr = o.universal_threshold(c[0, 0])
    srt = np.sort(np.abs(c[0, 0]))
    assert thr == ((srt[n // 2 - 1] + srt[n // 2]) / 2.0 / 0.6745) * math.sqrt(2 * math.log(n))
    d1 = o.apply_threshold(c[0, 0], thr, True)
    assert (d1 == 0).any()
    st, y1 = o.swt_denoise_batch(x, w, 4, -1.0, True, P)
    st, y2 = o.swt_denoise_batch(x, w, 4, -999.0, True, P)
    assert np.array_equal(y1, y2)
    assert np.sqrt(np.mean((y1 - x) ** 2)) < 0.5
    # odd length: single middle element
    assert o.estimate_noise_sigma(np.array([3.0, -1.0, 2.0])) == 2.0 / 0.6745


def test_extract_levels_sum_to_signal():
    # swt/VectorWaveSwtAdapterTest.java:205-217: sum of extractLevel(0..3) == signal at 1e-8
    w = get_wavelet("db4")
    x = signals(1, 128, seed=7)
    st, c = o.decompose_batch(x, w, 3, P)
    total = np.zeros_like(x)
    for keep in range(4):
        ck = np.zeros_like(c)
        ck[:, keep] = c[:, keep]
        total += o.reconstruct_batch(ck, w, P)
    np.testing.assert_allclose(total, x, atol=1e-8)


def test_c_oracle_matches_numpy_twin():
    """Independent statement (dilated form, no zero stuffing) of the same cascade in
    NumPy agrees bit for bit with the C restatement that zero-stuffs like the reference."""
    from oracle import oracle_np as onp
    for wn, n, j, bmode in [("db4", 100, 3, P), ("sym8", 777, 5, Z), ("coif3", 300, 3, S), ("bior4.4", 200, 3, P),
                            ("haar", 64, 5, P), ("db10", 1000, 4, Z), ("rbio1.3", 40, 4, P)]:
        w = get_wavelet(wn)
        x = signals(2, n, seed=n)
        st, c = o.decompose_batch(x, w, j, bmode)
        assert st == 0
        cn = onp.decompose(x, w, j, bmode)
        assert np.array_equal(c, cn), (wn, n, j, bmode)
        if len(w.rec_hi) >= len(w.rec_lo) or bmode == S:
            r = o.reconstruct_batch(c, w, bmode)
            rn = onp.reconstruct(c, w, bmode)
            assert np.array_equal(r, rn), (wn, n, j, bmode)
