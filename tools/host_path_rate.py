#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-pointer entry points (pageable host memory in/out)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import gpu_denoise, gpu_forward, gpu_inverse, signals  # noqa: E402
from vectorwave_amd import _native as N  # noqa: E402

ctx = N.Context(0)
b, n, j = 4096, 16384, 6
x = signals(b, n, seed=1)
gpu_forward(ctx, x[:64], "db4", j, 0)  # warm up
t0 = time.perf_counter(); st, c, _ = gpu_forward(ctx, x, "db4", j, 0); t1 = time.perf_counter()
st, xr, _ = gpu_inverse(ctx, c, "db4", 0); t2 = time.perf_counter()
st, y, _ = gpu_denoise(ctx, x, "db4", j, 0, -1.0, True); t3 = time.perf_counter()
out = {
    "workload": "%d x %d DB4 J=%d PERIODIC, pageable host buffers" % (b, n, j),
    "forward_Gsamples_per_s": b * n / (t1 - t0) / 1e9, "forward_host_GBps": (b * n * 8 * (j + 2)) / (t1 - t0) / 1e9,
    "inverse_Gsamples_per_s": b * n / (t2 - t1) / 1e9, "inverse_host_GBps": (b * n * 8 * (j + 2)) / (t2 - t1) / 1e9,
    "denoise_Gsamples_per_s": b * n / (t3 - t2) / 1e9, "denoise_host_GBps": (b * n * 8 * 2) / (t3 - t2) / 1e9,
}
# the same with page-locked buffers from vw_host_alloc (what a host that allocates through the library gets)
import ctypes as C
from vectorwave_amd.wavelets import get_wavelet
w = get_wavelet("db4")
dp = C.POINTER(C.c_double)
taps = [np.ascontiguousarray(t, dtype=np.float64) for t in (w.dec_lo, w.dec_hi, w.rec_lo, w.rec_hi)]
tp = [t.ctypes.data_as(dp) for t in taps]
xp = ctx.host_array((b, n)); xp[:] = x
cp = ctx.host_array((b, j + 1, n)); cp[:] = 0
yp = ctx.host_array((b, n)); yp[:] = 0
bad = C.c_int64(-1)
vp = lambda a: C.c_void_p(a.ctypes.data)
t0 = time.perf_counter()
assert ctx.lib.vw_modwt_forward_batch(ctx.handle, vp(xp), b, n, j, tp[0], 8, tp[1], 8, 0, 0, vp(cp), C.byref(bad)) == 0
t1 = time.perf_counter()
assert ctx.lib.vw_modwt_inverse_batch(ctx.handle, vp(cp), b, n, j, tp[2], 8, tp[3], 8, 0, 0, vp(yp), C.byref(bad)) == 0
t2 = time.perf_counter()
assert ctx.lib.vw_swt_denoise_batch(ctx.handle, vp(xp), b, n, j, tp[0], 8, tp[1], 8, tp[2], 8, tp[3], 8, 0, -1.0, -1.0, 1, vp(yp), None) == 0
t3 = time.perf_counter()
assert np.array_equal(cp, c) and np.array_equal(yp, y)
out["pinned"] = {
    "forward_Gsamples_per_s": b * n / (t1 - t0) / 1e9, "forward_host_GBps": (b * n * 8 * (j + 2)) / (t1 - t0) / 1e9,
    "inverse_Gsamples_per_s": b * n / (t2 - t1) / 1e9, "inverse_host_GBps": (b * n * 8 * (j + 2)) / (t2 - t1) / 1e9,
    "denoise_Gsamples_per_s": b * n / (t3 - t2) / 1e9, "denoise_host_GBps": (b * n * 8 * 2) / (t3 - t2) / 1e9,
}
print(json.dumps(out))
