"""Repeat the k_inv_wave6 partial-last-chunk case and print every differing element of the first run that differs.

usage: python tools/stress_wave_partial.py [n] [mode] [rows] [iterations] [runs_of]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from oracle import oracle as o  # noqa: E402
from util import gpu_inverse, signals  # noqa: E402
from vectorwave_amd import _native as N  # noqa: E402
from vectorwave_amd.wavelets import get_wavelet  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2306
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 600
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    runs_of = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    ctx = N.Context(0)
    ctx.set_plan_option("inv_wave", runs_of)
    w = get_wavelet("db4")
    x = signals(rows, n, seed=n)
    st, coeffs = o.decompose_batch(x, w, 6, mode)
    want = o.reconstruct_batch(coeffs, w, mode)
    wi = want.view(np.uint64)
    nbad = 0
    for it in range(iters):
        st, got, _ = gpu_inverse(ctx, coeffs, "db4", mode)
        assert st == 0, ctx.last_error()
        gi = got.view(np.uint64)
        if np.array_equal(gi, wi):
            continue
        nbad += 1
        idx = np.argwhere(gi != wi)
        print("iteration %d: %d elements differ (plan %s)" % (it, idx.shape[0], ctx.last_plan()), flush=True)
        for r, t in idx[:64]:
            print("  row %d t %d  got %s  want %s  coeffs a6 %s w1 %s" % (
                r, t, float(got[r, t]).hex(), float(want[r, t]).hex(), float(coeffs[r, 6, t]).hex(), float(coeffs[r, 0, t]).hex()), flush=True)
        if nbad >= 3:
            break
    print("done: %d of %d iterations differed" % (nbad, it + 1), flush=True)


if __name__ == "__main__":
    main()
