#!/usr/bin/env python3
"""Experiment: the long-row denoise (16384 x 16384, J = 6, UNIVERSAL) called in groups of G rows, one group after the
other on one stream, so that a group's coefficient block may stay in the last-level cache between the forward and the
inverse.  Prints ms for the whole batch per group size.

usage: python tools/exp_group_denoise.py [--rows 16384] [--n 16384] [--groups 0,64,128,256,512,1024,2048]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--levels", type=int, default=6)
    ap.add_argument("--groups", default="0,64,128,256,512,1024,2048")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--plan", default="")
    args = ap.parse_args()
    import torch
    from vectorwave_amd import _native as N
    from util import dev_signals, dev_wavelet_denoise

    ctx = N.Context(0)
    for kv in filter(None, args.plan.split(",")):
        k, v = kv.split("=")
        assert ctx.lib.vw_set_plan_option(ctx.handle, k.encode(), int(v)) == 0
    stream = torch.cuda.current_stream()
    x = dev_signals(args.rows, args.n, seed=6, noise=0.3)
    y = torch.empty_like(x)
    ref = None
    for g in [int(t) for t in args.groups.split(",")]:
        def run():
            if g == 0:
                dev_wavelet_denoise(ctx, x, "db4", args.levels, 0, 0, True, out=y)
            else:
                for r0 in range(0, args.rows, g):
                    dev_wavelet_denoise(ctx, x[r0:r0 + g], "db4", args.levels, 0, 0, True, out=y[r0:r0 + g])
        run()
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
        else:
            assert torch.equal(ref.view(torch.int64), y.view(torch.int64)), "grouped result differs"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.steps):
            run()
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        print("group %5d rows: %.3f ms  (%.1f Gsamples/s)  plan: %s" % (g, ms, args.rows * args.n / ms / 1e6, ctx.last_plan()), flush=True)


if __name__ == "__main__":
    main()
