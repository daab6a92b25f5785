#!/usr/bin/env python3
"""Experiment: per-call time of the streaming denoiser over ring lengths and estimators (run under rocprofv3
--kernel-trace to see k_stream_noise by itself; dispatch order = the order printed)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from vectorwave_amd import _native as N
from vectorwave_amd.wavelets import get_wavelet
from util import dev_signals

ctx = N.Context(0)
stream = torch.cuda.current_stream()
sp = C.c_void_p(stream.cuda_stream)
dp = C.POINTER(C.c_double)
w = get_wavelet("db4")
taps = [np.ascontiguousarray(t, dtype=np.float64) for t in (w.dec_lo, w.dec_hi, w.rec_lo, w.rec_hi)]
rows, n = 8192, 480
xs = [dev_signals(rows, n, seed=7 + q, noise=0.3) for q in range(8)]  # a new block per call: a ring of repeats is degenerate
x = xs[0]
y = torch.empty_like(x)
for window in (64, 256, 1024, 2048):
    for est, mult in ((0, 1.0), (1, 1.0)):
        state = C.c_void_p()
        assert ctx.lib.vw_stream_state_create(ctx.handle, rows, window, C.byref(state)) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(8):
            if it == 4:
                torch.cuda.synchronize()
                e0.record(stream)
            st = ctx.lib.vw_streaming_denoise_batch_dev(
                ctx.handle, state, C.c_void_p(xs[it % 8].data_ptr()), rows, n, taps[0].ctypes.data_as(dp), taps[0].size,
                taps[1].ctypes.data_as(dp), taps[1].size, taps[2].ctypes.data_as(dp), taps[2].size,
                taps[3].ctypes.data_as(dp), taps[3].size, 0, 0, 0, 1, mult, est, -1.0, -1.0, C.c_void_p(y.data_ptr()),
                None, None, 0, sp)
            assert st == 0
        e1.record(stream)
        torch.cuda.synchronize()
        print("window %d est %d: %.1f us/call" % (window, est, e0.elapsed_time(e1) / 4 * 1e3), flush=True)
        ctx.lib.vw_stream_state_destroy(ctx.handle, state)
