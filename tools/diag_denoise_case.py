"""One SWT-denoise case against the oracle with the plan printed, plus nearby variations (which factor matters).

usage: python tools/diag_denoise_case.py wavelet n levels rows mode seed
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

from oracle import oracle as o  # noqa: E402
from util import gpu_denoise, gpu_forward, gpu_inverse  # noqa: E402
from vectorwave_amd import _native as N  # noqa: E402
from vectorwave_amd.wavelets import get_wavelet  # noqa: E402


def ndiff(a, b):
    return int((np.ascontiguousarray(a).view(np.uint64) != np.ascontiguousarray(b).view(np.uint64)).sum())


def one(ctx, wn, n, j, b, mode, seed, thr, soft):
    w = get_wavelet(wn)
    rng = np.random.default_rng(seed)
    x = np.ascontiguousarray(np.sin(np.arange(n) / 5.0)[None] + 0.3 * rng.standard_normal((b, n)))
    st, want = o.swt_denoise_batch(x, w, j, thr, soft, mode)
    if st != 0:
        print("  %s n=%d J=%d mode=%d: oracle status %d" % (wn, n, j, mode, st))
        return
    gst, got, gthr = gpu_denoise(ctx, x, wn, j, mode, thr, soft)
    plan = ctx.last_plan()
    st, c = o.decompose_batch(x, w, j, mode)
    gst2, gc, _ = gpu_forward(ctx, x, wn, j, mode)
    fplan = ctx.last_plan()
    gst3, gx, _ = gpu_inverse(ctx, c, wn, mode)
    iplan = ctx.last_plan()
    print("  %s n=%d J=%d b=%d mode=%d thr=%g soft=%d: denoise status %d differ %d of %d | fwd differ %d | inv differ %d" % (
        wn, n, j, b, mode, thr, soft, gst, ndiff(got, want), got.size, ndiff(gc, c), ndiff(gx, o.reconstruct_batch(c, w, mode))))
    print("      denoise plan: %s\n      forward plan: %s\n      inverse plan: %s" % (plan, fplan, iplan))
    if ndiff(got, want):
        bands = [ndiff(gc[:, k], c[:, k]) for k in range(j + 1)]
        print("      forward bands differing: %s; thresholds %s" % (bands, gthr[:2]))


def main():
    wn, n, j, b, mode, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    ctx = N.Context(0)
    soft = bool(seed & 1)
    thr = -1.0 if seed & 2 else 0.2
    print("the case:")
    one(ctx, wn, n, j, b, mode, seed, thr, soft)
    print("variations:")
    for (vn, vj, vm, vthr, vsoft) in [(n, j - 1, mode, thr, soft), (n + 1, j, mode, thr, soft), (n - 1, j, mode, thr, soft), (n, j, 1, thr, soft),
                                      (n, j, mode, -1.0, soft), (n, j, mode, thr, not soft), (n, j - 2, mode, thr, soft), (n, 8, mode, thr, soft),
                                      (n, 6, mode, thr, soft)]:
        one(ctx, wn, vn, vj, b, vm, seed, vthr, vsoft)
    one(ctx, wn, n, j, 1, mode, seed, thr, soft)


if __name__ == "__main__":
    main()
