"""Every kernel family on batches that keep the whole chip busy, every row compared with the CPU oracle bit for bit.

The per-family parity tests use two or three rows so that the oracle stays cheap; a handful of workgroups never
overlap in the ways a full chip makes them overlap -- workgroups dispatched into LDS that others have just left, waves
of one CU in every phase of their pipelines at once, counted memory waits that are only tight when the memory system is
busy.  Two wrong-result races of the partial last chunk of k_inv_wave6 showed up only with hundreds of rows.  Here each
family runs 400-700 ragged rows (last tile partial, odd and even lengths), several workgroups per CU and several rounds
of dispatch.  Reference semantics as in the per-family tests (modwt/MultiLevelMODWTTransform.java:200-353, 362-468,
621-689; swt/VectorWaveSwtAdapter.java:296-312)."""
import numpy as np
import pytest

from oracle import oracle as o
from util import assert_bits_equal, gpu_denoise, gpu_forward, gpu_inverse, signals
from vectorwave_amd.wavelets import get_wavelet

pytestmark = pytest.mark.gpu

P, Z, S = 0, 1, 2


def _options(plan, options):
    for name, value in options:
        plan(name, value)


# (id, wavelet, n, levels, rows, mode, plan options, kernel expected in the forward plan)
FWD = [
    ("stream-runs1", "db4", 6002, 6, 600, P, [("stream_wgs", 1), ("fwd_chunks", 1)], "k_fwd_stream"),
    ("stream-runs3", "db4", 6002, 6, 600, Z, [("stream_wgs", 1), ("fwd_chunks", 3)], "k_fwd_stream"),
    ("stream-default", "db4", 5000, 6, 700, P, [], "k_fwd_stream"),
    ("stream-odd-n", "db4", 5121, 5, 500, P, [], "k_fwd_stream"),
    ("stream-haar-J8", "haar", 5000, 8, 600, Z, [], "k_fwd_stream"),
    ("stream-sym", "db4", 5000, 6, 600, S, [], "k_fwd_stream"),
    ("sdry", "sym8", 9000, 7, 300, P, [], "k_fwd_sdry"),
    ("sdry-runs3", "sym8", 9000, 7, 300, Z, [("sdry_chunks", 3)], "k_fwd_sdry"),
    ("sdry-db10", "db10", 7001, 6, 300, P, [], "k_fwd_sdry"),
]


@pytest.mark.parametrize("name,wn,n,j,b,mode,options,kernel", FWD, ids=[c[0] for c in FWD])
def test_forward_families_under_load(ctx, plan, name, wn, n, j, b, mode, options, kernel):
    _options(plan, options)
    w = get_wavelet(wn)
    x = signals(b, n, seed=n + j)
    st, want = o.decompose_batch(x, w, j, mode)
    assert st == 0
    for _ in range(2):
        st, got, _bad = gpu_forward(ctx, x, wn, j, mode)
        assert st == 0, ctx.last_error()
        assert kernel in ctx.last_plan(), ctx.last_plan()
        assert_bits_equal(got, want, "forward %s: %s N=%d J=%d, %d rows" % (name, wn, n, j, b))


# (id, wavelet, n, levels, rows, mode, plan options, kernel expected in the inverse plan)
INV = [
    ("ring", "db4", 6002, 6, 600, P, [], "k_inv_ring"),
    ("ring-zero", "db4", 4610, 6, 700, Z, [], "k_inv_ring"),
    ("ring-haar-J8", "haar", 5000, 8, 600, P, [], "k_inv_ring"),
    ("carry2", "db4", 6002, 6, 600, P, [("carry_chunks", 2)], "k_inv_carry"),
    ("carry3", "db4", 8194, 6, 500, Z, [("carry_chunks", 3)], "k_inv_carry"),
    ("blk", "db4", 6002, 6, 600, Z, [("no_ring", 1)], "k_inv_blk"),
    ("blk-odd-n", "db4", 5121, 5, 500, P, [], "k_inv_blk"),
    ("fused", "db4", 6002, 6, 600, P, [("no_blk", 1)], "k_inv_fused"),
    ("wave-runs2", "db4", 4738, 6, 700, P, [("inv_wave", 2)], "k_inv_wave"),
    ("wave-runs3", "db4", 4610, 6, 700, Z, [("inv_wave", 3)], "k_inv_wave"),
    ("wave-whole-rows", "db4", 2562, 6, 700, P, [("inv_wave", 1000)], "k_inv_wave"),
    ("wave-haar", "haar", 3202, 6, 700, P, [("inv_wave", 4)], "k_inv_wave"),
    ("wave-db2", "db2", 3330, 6, 700, Z, [("inv_wave", 2)], "k_inv_wave"),
    ("sym", "db4", 5000, 6, 600, S, [], "k_inv_sym"),
    ("sym-odd-n", "db4", 4097, 5, 500, S, [], "k_inv_sym"),
    ("sdry", "sym8", 9000, 7, 300, P, [], "k_inv_sdry"),
    ("sdry-runs3", "sym8", 9000, 7, 300, Z, [("sdry_chunks", 3)], "k_inv_sdry"),
    ("sdry-db10", "db10", 7001, 6, 300, P, [], "k_inv_sdry"),
]


@pytest.mark.parametrize("name,wn,n,j,b,mode,options,kernel", INV, ids=[c[0] for c in INV])
def test_inverse_families_under_load(ctx, plan, name, wn, n, j, b, mode, options, kernel):
    _options(plan, options)
    w = get_wavelet(wn)
    x = signals(b, n, seed=n + 2 * j)
    st, coeffs = o.decompose_batch(x, w, j, mode)
    assert st == 0
    want = o.reconstruct_batch(coeffs, w, mode)
    for _ in range(2):
        st, got, _bad = gpu_inverse(ctx, coeffs, wn, mode)
        assert st == 0, ctx.last_error()
        assert kernel in ctx.last_plan(), ctx.last_plan()
        assert_bits_equal(got, want, "inverse %s: %s N=%d J=%d, %d rows" % (name, wn, n, j, b))


# (id, wavelet, n, levels, rows, mode, threshold (< 0: universal), soft, plan options)
DEN = [
    ("ring-universal-soft", "db4", 6002, 6, 500, P, -1.0, True, []),
    ("ring-fixed-hard", "db4", 4610, 6, 600, Z, 0.25, False, []),
    ("wave-universal-soft", "db4", 4738, 6, 600, P, -1.0, True, [("inv_wave", 3)]),
    ("wave-universal-hard", "db4", 4610, 6, 600, Z, -1.0, False, [("inv_wave", 2)]),
    ("sym-universal-soft", "db4", 5000, 6, 500, S, -1.0, True, []),
    ("sdry-universal-soft", "sym8", 9000, 7, 250, P, -1.0, True, []),
    ("rows-in-one-launch", "db4", 480, 4, 4000, P, -1.0, True, []),
    ("rows-in-one-launch-zero", "db4", 1000, 5, 2000, Z, -1.0, False, []),
    ("rows-of-2048", "haar", 2048, 6, 1200, P, -1.0, True, []),
]


@pytest.mark.parametrize("name,wn,n,j,b,mode,thr,soft,options", DEN, ids=[c[0] for c in DEN])
def test_denoise_families_under_load(ctx, plan, name, wn, n, j, b, mode, thr, soft, options):
    _options(plan, options)
    w = get_wavelet(wn)
    x = signals(b, n, seed=n + 5 * j)
    st, want = o.swt_denoise_batch(x, w, j, thr, soft, mode)
    assert st == 0
    for _ in range(2):
        st, got, _thr = gpu_denoise(ctx, x, wn, j, mode, thr, soft)
        assert st == 0, ctx.last_error()
        assert_bits_equal(got, want, "denoise %s: %s N=%d J=%d, %d rows" % (name, wn, n, j, b))
        assert np.isfinite(got).all()
