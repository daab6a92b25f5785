#!/usr/bin/env python3
"""Would the long-row denoise gain from running the median of one row group beside the transforms of another?

The denoise of 16384 x 16384 (J = 6) is forward (memory-bound) -> exact median of |d_1| (latency-bound selection) ->
thresholding inverse (memory-bound).  This experiment splits the rows into G groups and enqueues, on two streams,
    stream A: forward(g0), forward(g1), ..., then inverse(g0), inverse(g1), ...
    stream B: median(g0) after forward(g0), median(g1) after forward(g1), ...
with events for the dependencies, through the public device entry points (forward, universal threshold; the
thresholding inverse is stood in for by the plain inverse, which costs the same within a few percent), and compares
the wall time with the same kernels enqueued on one stream.
usage: tools/exp_overlap_denoise.py [rows n levels]"""
import ctypes as C
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from vectorwave_amd import _native as N
    from vectorwave_amd.wavelets import get_wavelet
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    J = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    dev = torch.device("cuda", 0)
    ctx = N.Context(0)
    w = get_wavelet("db4")
    dp = C.POINTER(C.c_double)
    h, g, hr, gr = (np.ascontiguousarray(v, dtype=np.float64) for v in (w.dec_lo, w.dec_hi, w.rec_lo, w.rec_hi))
    x = torch.randn((rows, n), dtype=torch.float64, device=dev)
    coeffs = torch.empty((rows, J + 1, n), dtype=torch.float64, device=dev)
    y = torch.empty((rows, n), dtype=torch.float64, device=dev)
    thr = torch.empty((rows,), dtype=torch.float64, device=dev)
    s2 = math.sqrt(2 * math.log(n))
    lib, hnd = ctx.lib, ctx.handle

    def fwd(r0, r1, stream):
        st = lib.vw_modwt_forward_batch_dev(hnd, C.c_void_p(x[r0:r1].data_ptr()), r1 - r0, n, J, h.ctypes.data_as(dp), h.size,
                                            g.ctypes.data_as(dp), g.size, 0, 0, C.c_void_p(coeffs[r0:r1].data_ptr()), None,
                                            C.c_void_p(stream.cuda_stream))
        assert st == 0, ctx.last_error()

    def med(r0, r1, stream):
        st = lib.vw_swt_universal_threshold_batch_dev(hnd, C.c_void_p(coeffs[r0:r1].data_ptr()), r1 - r0, n, J, s2,
                                                      C.c_void_p(thr[r0:r1].data_ptr()), C.c_void_p(stream.cuda_stream))
        assert st == 0, ctx.last_error()

    def inv(r0, r1, stream):
        st = lib.vw_modwt_inverse_batch_dev(hnd, C.c_void_p(coeffs[r0:r1].data_ptr()), r1 - r0, n, J, hr.ctypes.data_as(dp), hr.size,
                                            gr.ctypes.data_as(dp), gr.size, 0, 0, C.c_void_p(y[r0:r1].data_ptr()), None,
                                            C.c_void_p(stream.cuda_stream))
        assert st == 0, ctx.last_error()

    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def serial(groups):
        for r0, r1 in groups:
            fwd(r0, r1, sa)
        for r0, r1 in groups:
            med(r0, r1, sa)
        for r0, r1 in groups:
            inv(r0, r1, sa)

    def overlapped(groups):
        done_f = [torch.cuda.Event() for _ in groups]
        done_m = [torch.cuda.Event() for _ in groups]
        for k, (r0, r1) in enumerate(groups):
            fwd(r0, r1, sa)
            done_f[k].record(sa)
            sb.wait_event(done_f[k])
            med(r0, r1, sb)
            done_m[k].record(sb)
        for k, (r0, r1) in enumerate(groups):
            sa.wait_event(done_m[k])
            inv(r0, r1, sa)

    def timed(fn, groups, reps=6):
        for _ in range(2):
            fn(groups)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(sa)
        for _ in range(reps):
            fn(groups)
            sa.wait_stream(sb)
        e1.record(sa)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for G in (1, 2, 4, 8, 16):
        step = (rows + G - 1) // G
        groups = [(r0, min(rows, r0 + step)) for r0 in range(0, rows, step)]
        ts = timed(serial, groups)
        to = timed(overlapped, groups)
        print("groups %2d: one stream %.3f ms, two streams %.3f ms" % (G, ts, to), flush=True)


if __name__ == "__main__":
    main()
