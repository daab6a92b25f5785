#!/usr/bin/env python3
"""A/B of equivalent launch plans inside ONE process on one GPU: the batch is allocated once and the variants
(sets of vw_set_plan_option settings) are timed in alternation, so box-to-box and run-to-run drift cancels.
usage: tools/ab_plan.py [--rows R --n N --levels J --wavelet W --boundary B --rounds K] name:opt=val,opt=val ...
       e.g. tools/ab_plan.py base: noring:no_ring=1 old:no_blk=1"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--levels", type=int, default=6)
    ap.add_argument("--wavelet", default="db4")
    ap.add_argument("--boundary", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--fma", action="store_true")
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()
    import torch
    from vectorwave_amd import _native as N
    from vectorwave_amd.wavelets import get_wavelet
    dev = torch.device("cuda", 0)
    ctx = N.Context(0)
    w = get_wavelet(args.wavelet)
    dp = C.POINTER(C.c_double)
    h, g, hr, gr = (np.ascontiguousarray(v, dtype=np.float64) for v in (w.dec_lo, w.dec_hi, w.rec_lo, w.rec_hi))
    rows, n, J = args.rows, args.n, args.levels
    x = torch.randn((rows, n), dtype=torch.float64, device=dev)
    coeffs = torch.empty((rows, J + 1, n), dtype=torch.float64, device=dev)
    xr = torch.empty((rows, n), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)
    flags = N.VW_FLAG_FMA if args.fma else 0

    def fwd():
        st = ctx.lib.vw_modwt_forward_batch_dev(ctx.handle, C.c_void_p(x.data_ptr()), rows, n, J, h.ctypes.data_as(dp), h.size,
                                                g.ctypes.data_as(dp), g.size, args.boundary, flags, C.c_void_p(coeffs.data_ptr()), None, sp)
        assert st == 0, ctx.last_error()

    def inv():
        st = ctx.lib.vw_modwt_inverse_batch_dev(ctx.handle, C.c_void_p(coeffs.data_ptr()), rows, n, J, hr.ctypes.data_as(dp), hr.size,
                                                gr.ctypes.data_as(dp), gr.size, args.boundary, flags, C.c_void_p(xr.data_ptr()), None, sp)
        assert st == 0, ctx.last_error()

    variants = []
    for v in args.variants:
        name, _, opts = v.partition(":")
        variants.append((name, [tuple(o.split("=")) for o in opts.split(",") if o]))
    res = {name: {"fwd": [], "inv": []} for name, _ in variants}
    ref = None
    for rnd in range(args.rounds):
        for name, opts in variants:
            ctx.set_plan_option("reset", 0)
            for k, v in opts:
                ctx.set_plan_option(k, int(v))
            fwd(); inv()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record(stream)
            for _ in range(args.reps):
                fwd()
            e[1].record(stream)
            for _ in range(args.reps):
                inv()
            e[2].record(stream)
            torch.cuda.synchronize()
            res[name]["fwd"].append(e[0].elapsed_time(e[1]) / args.reps)
            res[name]["inv"].append(e[1].elapsed_time(e[2]) / args.reps)
            if rnd == 0:  # all variants must agree bit for bit
                cur = (coeffs[:4].cpu().numpy().view(np.uint64).copy(), xr[:4].cpu().numpy().view(np.uint64).copy())
                if ref is None:
                    ref = cur
                else:
                    assert np.array_equal(ref[0], cur[0]) and np.array_equal(ref[1], cur[1]), "variant %s differs" % name
    for name, _ in variants:
        f, i = np.array(res[name]["fwd"]), np.array(res[name]["inv"])
        print("%-12s fwd med %.3f min %.3f | inv med %.3f min %.3f | fwd+inv %.2f Gs/s" % (
            name, np.median(f), f.min(), np.median(i), i.min(), rows * n / ((np.median(f) + np.median(i)) * 1e-3) / 1e9), flush=True)


if __name__ == "__main__":
    main()
