#!/usr/bin/env python3
"""Experiment: a few calls of the streaming denoiser at one configuration (for counter passes).
usage: exp_stream_one.py <window> <estimation> <multiplier> [rows] [n]"""
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

window, est, mult = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
n = int(sys.argv[5]) if len(sys.argv) > 5 else 480
ctx = N.Context(0)
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
dp = C.POINTER(C.c_double)
w = get_wavelet("db4")
taps = [np.ascontiguousarray(t, dtype=np.float64) for t in (w.dec_lo, w.dec_hi, w.rec_lo, w.rec_hi)]
xs = [dev_signals(rows, n, seed=7 + q, noise=0.3) for q in range(8)]  # a new block per call: a ring of repeats is degenerate
x = xs[0]
y = torch.empty_like(x)
state = C.c_void_p()
assert ctx.lib.vw_stream_state_create(ctx.handle, rows, window, C.byref(state)) == 0
for it in range(6):
    st = ctx.lib.vw_streaming_denoise_batch_dev(
        ctx.handle, state, C.c_void_p(xs[it % 8].data_ptr()), rows, n, taps[0].ctypes.data_as(dp), taps[0].size,
        taps[1].ctypes.data_as(dp), taps[1].size, taps[2].ctypes.data_as(dp), taps[2].size,
        taps[3].ctypes.data_as(dp), taps[3].size, 0, 0, 0, 1, mult, est, -1.0, -1.0, C.c_void_p(y.data_ptr()),
        None, None, 0, sp)
    assert st == 0
torch.cuda.synchronize()
ctx.lib.vw_stream_state_destroy(ctx.handle, state)
