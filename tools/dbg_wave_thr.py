import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import oracle as o
from util import gpu_denoise, signals
from vectorwave_amd import _native as N
from vectorwave_amd.wavelets import get_wavelet
ctx = N.Context(0)
w = get_wavelet("db4")
x = signals(2, 5000, seed=17)
for soft in (True, False):
    for thr in (0.0, 0.25, 1e9, -1.0):
        for nch in (-1, 1000):
            ctx.set_plan_option("reset", 0); ctx.set_plan_option("inv_wave", nch)
            st, want = o.swt_denoise_batch(x, w, 6, thr, soft, 0)
            st, got, _ = gpu_denoise(ctx, x, "db4", 6, 0, thr, soft)
            bad = (got.view(np.uint64) != want.view(np.uint64))
            print("soft", soft, "thr", thr, "nch", nch, ctx.last_plan().split(";")[1], "mismatches", bad.sum(), "first", np.argwhere(bad)[:1].tolist(), "maxdiff", np.abs(got-want).max())
ctx.set_plan_option("reset", 0); ctx.set_plan_option("inv_wave", 1000)
for thr in (1e9,):
    st, want = o.swt_denoise_batch(x, w, 6, thr, False, 0)
    st, got, _ = gpu_denoise(ctx, x, "db4", 6, 0, thr, False)
    bad = (got.view(np.uint64) != want.view(np.uint64))
    for r in range(2):
        idx = np.nonzero(bad[r])[0]
        print("row", r, "bad", idx.size, "per chunk", np.bincount(idx // 512, minlength=10).tolist(), "first offsets", (idx[:8] % 512).tolist())
# the same through the device entry point with explicit per-call checks of the coefficient thresholds
