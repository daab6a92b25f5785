#!/usr/bin/env python3
"""Freeze golden vectors for the MODWT/SWT path under tests/golden/.

The reference is Java and cannot run in the build image, and its tests hold no
numeric golden files -- only closed-form known answers.  So two fixtures are kept:

  reference_kats.json   the closed-form known-answer vectors transcribed (as data:
                        inputs and expected outputs) from the reference's unit tests,
                        with the test file:line of each
  oracle_vectors.npz    seeded inputs and the outputs of the CPU oracle (the C restatement
                        of the reference's algorithm, itself pinned by the KATs) for every
                        (wavelet, boundary, N, J) case listed below: coefficient blocks,
                        reconstructions, denoise outputs and thresholds

Run from the repo root:  python tools/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as o  # noqa: E402
from util import signals  # noqa: E402
from vectorwave_amd.wavelets import get_wavelet  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

KATS = {
    "_comment": "Closed-form known answers held by the reference's unit tests (paths under "
                "src/test/java/ai/prophetizo/wavelet/). Data only.",
    "haar_level1_approx": {
        "source": "modwt/MODWTPercivalWaldenValidationTest.java:73-82",
        "wavelet": "haar", "boundary": 0, "signal": [1.0, 2.0, 3.0, 4.0],
        "approx": [2.5, 1.5, 2.5, 3.5], "tol": 1e-12},
    "circular_06_04": {
        "source": "internal/ScalarOpsTest.java:225-240",
        "signal": [1.0, 2.0, 3.0, 4.0], "filter": [0.6, 0.4], "boundary": 0,
        "expected": [2.2, 1.6, 2.6, 3.6], "tol": 1e-10},
    "zero_pad_06_04": {
        "source": "internal/ScalarOpsTest.java:262-274",
        "signal": [1.0, 2.0, 3.0, 4.0], "filter": [0.6, 0.4], "boundary": 1,
        "expected": [0.6, 1.6, 2.6, 3.6], "tol": 1e-10},
    "level2_dilation": {
        "source": "internal/ScalarOpsTest.java:245-258",
        "signal": [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0], "filter": [0.7, 0.3], "level": 2,
        "expected_first_two": [2.8, 3.8], "tol": 1e-10},
    "time_reversed_haar": {
        "source": "modwt/TimeReversedFilterTest.java:20-50",
        "signal": [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0], "filter": [0.7071067811865475, 0.7071067811865475],
        "expected_first_two": [0.7071067811865475 * 1 + 0.7071067811865475 * 8,
                               0.7071067811865475 * 2 + 0.7071067811865475 * 1], "tol": 1e-10},
    "manual_circular": {
        "source": "modwt/MODWTMathematicalValidationTest.java:296-316",
        "signal": [1.0, 4.0, 2.0, 8.0], "filter": [0.5, -0.5], "boundary": 0,
        "expected": [-3.5, 1.5, -1.0, 3.0], "tol": 1e-12},
    "max_levels": {
        "source": "modwt/MultiLevelMODWTTransformTest.java:254-291",
        "cases": [[10000, 2, 9], [8, 8, 0], [9, 8, 1]]},
    "soa_layout": {
        "source": "modwt/BatchSIMDMODWTTest.java:28-52",
        "rows": [[1.0, 2.0, 3.0, 4.0], [5.0, 6.0, 7.0, 8.0], [9.0, 10.0, 11.0, 12.0]],
        "soa": [1.0, 5.0, 9.0, 2.0, 6.0, 10.0, 3.0, 7.0, 11.0, 4.0, 8.0, 12.0]},
}

CASES = [  # wavelet, N, J
    ("haar", 64, 3), ("haar", 8, 3), ("db2", 64, 3), ("db4", 64, 3), ("db4", 301, 5), ("db4", 480, 4),
    ("db10", 400, 4), ("sym8", 333, 4), ("coif3", 301, 4), ("bior4.4", 200, 4),
]


def main():
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "reference_kats.json"), "w") as f:
        json.dump(KATS, f, indent=1)
    arrays = {}
    names = []
    for wn, n, j in CASES:
        w = get_wavelet(wn)
        x = signals(2, n, seed=1000 + n + j)
        for mode in (0, 1, 2):
            key = "%s_N%d_J%d_m%d" % (wn.replace(".", "_"), n, j, mode)
            st, c = o.decompose_batch(x, w, j, mode)
            assert st == 0, key
            arrays[key + "_x"] = x
            arrays[key + "_coeffs"] = c
            arrays[key + "_recon"] = o.reconstruct_batch(c, w, mode)
            if mode != 2:
                st, y = o.swt_denoise_batch(x, w, j, -1.0, True, mode)
                arrays[key + "_denoise_soft"] = y
                arrays[key + "_thr"] = np.array([o.universal_threshold(c[i, 0]) for i in range(2)])
            names.append(key)
        c1 = o.modwt_forward_batch(x, w, 0)
        arrays["%s_N%d_single_coeffs" % (wn.replace(".", "_"), n)] = c1
    arrays["_cases"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "oracle_vectors.npz"), **arrays)
    sz = os.path.getsize(os.path.join(OUT, "oracle_vectors.npz"))
    print("wrote %d cases, %.1f KiB" % (len(names), sz / 1024))
    make_extra()
    make_streaming()
    make_seeded()


# second fixture: the WaveletDenoiser path and level groups with a long context (more than one
# 512-sample chunk of history), which run different kernels from the cases above
DENOISER_CASES = [  # wavelet, N, levels (0 = single-level form), method, soft, boundary
    ("haar", 64, 0, 3, 1, 0), ("db4", 250, 0, 1, 0, 0), ("db4", 250, 0, 4, 1, 1), ("db4", 512, 4, 0, 1, 0),
    ("db4", 512, 4, 1, 1, 0), ("db4", 512, 4, 2, 0, 1), ("db4", 512, 4, 3, 1, 2), ("sym8", 700, 3, 3, 0, 0),
]
LONG_CONTEXT_CASES = [("sym8", 1300, 6), ("db10", 1001, 5)]  # wavelet, N, J


def make_extra():
    arrays, dn, lc = {}, [], []
    for wn, n, levels, method, soft, mode in DENOISER_CASES:
        w = get_wavelet(wn)
        x = signals(2, n, seed=2000 + n + levels + method)
        key = "%s_N%d_L%d_M%d_S%d_m%d" % (wn.replace(".", "_"), n, levels, method, soft, mode)
        st, y, thr = o.wavelet_denoise_batch(x, w, levels, method, bool(soft), mode, fixed_threshold=0.2)
        assert st == 0, key
        arrays[key + "_x"], arrays[key + "_y"], arrays[key + "_thr"] = x, y, thr
        dn.append(key)
    for wn, n, j in LONG_CONTEXT_CASES:
        w = get_wavelet(wn)
        x = signals(1, n, seed=3000 + n + j)
        for mode in (0, 1):
            key = "%s_N%d_J%d_m%d" % (wn.replace(".", "_"), n, j, mode)
            st, c = o.decompose_batch(x, w, j, mode)
            assert st == 0, key
            arrays[key + "_x"], arrays[key + "_coeffs"] = x, c
            arrays[key + "_recon"] = o.reconstruct_batch(c, w, mode)
            lc.append(key)
    arrays["_denoiser_cases"] = np.array(dn)
    arrays["_long_context_cases"] = np.array(lc)
    path = os.path.join(OUT, "denoiser_longcontext_vectors.npz")
    np.savez_compressed(path, **arrays)
    print("wrote %d + %d extra cases, %.1f KiB" % (len(dn), len(lc), os.path.getsize(path) / 1024))


# third fixture: the streaming callers (block sequences with their state) and the sparse SWT result
STREAMING_CASES = [  # wavelet, block length, ring length, estimation, method, multiplier, soft, boundary
    ("db4", 96, 256, 0, 0, 1.0, 1, 0), ("db4", 96, 64, 0, 0, 1.5, 1, 0), ("haar", 50, 128, 1, 2, 0.7, 0, 1),
    ("sym8", 120, 100, 1, 1, 1.0, 1, 2), ("db4", 80, 32, 2, 4, 2.0, 1, 0), ("db4", 300, 10, 0, 1, 1.2, 0, 0),
]


def make_streaming():
    arrays, names = {}, []
    for wn, n, ring, est, method, mult, soft, mode in STREAMING_CASES:
        w = get_wavelet(wn)
        key = "%s_n%d_w%d_e%d_M%d_x%g_S%d_m%d" % (wn, n, ring, est, method, mult, soft, mode)
        state = o.StreamingDenoiserState(2, ring)
        rng = np.random.default_rng(4000 + n + ring)
        blocks = np.sin(np.arange(4 * n).reshape(4, 1, n) / 7.0) + 0.2 * rng.standard_normal((4, 2, n))
        outs, levels, rings = [], [], []
        for blk in blocks:
            st, y, _ = o.streaming_denoise_batch(state, blk, w, method, bool(soft), mult, est, mode)
            assert st == 0, key
            outs.append(y)
            levels.append(state.level.copy())
            rings.append(state.window.copy())
        arrays[key + "_x"], arrays[key + "_y"] = blocks, np.array(outs)
        arrays[key + "_level"], arrays[key + "_ring"] = np.array(levels), np.array(rings)
        names.append(key)
    # block transforms over one stream fed in uneven pieces
    x = signals(1, 1000, seed=4100)[0]
    w = get_wavelet("db4")
    t = o.StreamingTransformState(w, 128, 0)
    arrays["xform_x"] = x
    arrays["xform_blocks"] = np.concatenate([t.process(x), t.flush()])
    st, ml = o.stream_multilevel(x, w, 128, 3, 0)
    arrays["xform_multilevel_blocks"] = ml
    # sparse detail band
    d = signals(1, 500, seed=4200)[0] - 0.3
    idx, val = o.sparse_vector(d, 0.6)
    arrays["sparse_dense"], arrays["sparse_idx"], arrays["sparse_val"] = d, idx, val
    arrays["_streaming_cases"] = np.array(names)
    path = os.path.join(OUT, "streaming_sparse_vectors.npz")
    np.savez_compressed(path, **arrays)
    print("wrote %d streaming cases, %.1f KiB" % (len(names), os.path.getsize(path) / 1024))


# fourth fixture: the reference tests' own seeded inputs.  Those tests draw from java.util.Random(42).nextGaussian();
# tools/java_random.py restates that generator (48-bit LCG, polar method, fdlibm log) so the inputs are the reference's
# bit for bit.  What the reference asserts on them are properties (batch == per-signal at 1e-10, parallel ==
# sequential, "zeroes some, keeps some", shapes); the fixture keeps the inputs and the oracle's outputs on them.
def make_seeded():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from java_random import JavaRandom
    arrays = {}
    db4, haar = get_wavelet("db4"), get_wavelet("haar")
    # modwt/BatchSIMDMODWTTest.java:104-151: 16 signals x 64 samples, Random(42), row by row
    r = JavaRandom(42)
    xb = np.array([[r.next_gaussian() for _ in range(64)] for _ in range(16)])
    arrays["batch_db4_x"] = xb
    arrays["batch_db4_coeffs"] = o.modwt_forward_batch(xb, db4, 0)
    a, d = o.batch_modwt_soa(o.convert_to_soa(xb), db4, 16, 64)
    arrays["batch_db4_soa_approx"], arrays["batch_db4_soa_detail"] = a, d
    # modwt/BatchSIMDMODWTTest.java:153-183: 7 signals x 32 samples, b + t * 0.1 (exact arithmetic)
    x7 = np.array([[b + t * 0.1 for t in range(32)] for b in range(7)])
    arrays["batch_haar7_x"] = x7
    a, d = o.batch_modwt_soa(o.convert_to_soa(x7), haar, 7, 32)
    arrays["batch_haar7_soa_approx"], arrays["batch_haar7_soa_detail"] = a, d
    # modwt/ParallelMultiLevelMODWTTest.java:64-100: 128 samples, Random(42), DB4, 4 levels, PERIODIC
    r = JavaRandom(42)
    xp = np.array([r.next_gaussian() for _ in range(128)])
    arrays["parallel_db4_x"] = xp
    st, c = o.decompose_batch(xp[None], db4, 4, 0)
    assert st == 0
    arrays["parallel_db4_coeffs"] = c[0]
    st, cp = o.parallel_decompose(xp, db4, 4, 0)
    assert st == 0
    arrays["parallel_db4_parallel_coeffs"] = cp
    # swt/VectorWaveSwtAdapterTest.java:243-268 with :423-439: sin(2 pi i / 16) + 0.5 cos(4 pi i / 16) + 0.1 * gaussian,
    # Random(42), DB4, 4 levels, universal soft threshold.  (Math.sin / Math.cos of the JVM and of the C library may
    # differ in the last bit, so this input is the reference's only up to that.)
    r = JavaRandom(42)
    i = np.arange(256)
    clean = np.sin(2 * np.pi * i / 16.0) + 0.5 * np.cos(4 * np.pi * i / 16.0)
    xs = np.array([clean[k] + 0.1 * r.next_gaussian() for k in range(256)])
    arrays["swt_universal_x"] = xs
    st, cs = o.decompose_batch(xs[None], db4, 4, 0)
    assert st == 0
    thr = o.universal_threshold(cs[0, 0])
    arrays["swt_universal_thr"] = np.array([thr])
    ct = cs.copy()
    for lev in range(4):
        ct[0, lev] = o.apply_threshold(cs[0, lev], thr, True)
    arrays["swt_universal_thresholded"] = ct[0]
    st, y = o.swt_denoise_batch(xs[None], db4, 4, -1.0, True, 0)
    assert st == 0
    arrays["swt_universal_denoised"] = y[0]
    # integration/SimplifiedAPIWorkflowTest.java:103-144 with :231-237: 32 sine waves of 777 samples, DB4 PERIODIC
    # forwardBatch, softThreshold(detail, 0.1), inverseBatch (same remark on Math.sin)
    xw = np.array([[1.0 * np.sin(2 * np.pi * (0.01 * (k + 1)) * t) for t in range(777)] for k in range(32)])
    arrays["workflow_x"] = xw
    cw = o.modwt_forward_batch(xw, db4, 0)
    arrays["workflow_coeffs"] = cw
    cm = cw.copy()
    for k in range(32):
        cm[k, 0] = o.soft_threshold(cw[k, 0], 0.1)
    arrays["workflow_reconstructed"] = o.modwt_inverse_batch(cm, db4, 0)
    path = os.path.join(OUT, "reference_seeded.npz")
    np.savez_compressed(path, **arrays)
    print("wrote the reference tests' seeded inputs, %.1f KiB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "seeded":
        make_seeded()
    else:
        main()
