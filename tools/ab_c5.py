#!/usr/bin/env python3
"""In-process timing of the SWT denoise (BASELINE config 5 shape by default) under plan options.
usage: tools/ab_c5.py [--rows 8192 --n 480 --levels 4 --wavelet db4] name:opt=val,... ..."""
import argparse, ctypes as C, math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8192)
    ap.add_argument("--n", type=int, default=480)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--wavelet", default="db4")
    ap.add_argument("--boundary", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()
    import torch
    from util import dev_denoise, dev_signals
    from vectorwave_amd import _native as N
    ctx = N.Context(0)
    x = dev_signals(args.rows, args.n, seed=1, noise=0.5)
    y = torch.empty_like(x)
    stream = torch.cuda.current_stream()
    res, ref = {}, None
    for rnd in range(args.rounds):
        for v in args.variants:
            name, _, opts = v.partition(":")
            ctx.set_plan_option("reset", 0)
            for o in [o for o in opts.split(",") if o]:
                k, val = o.split("=")
                ctx.set_plan_option(k, int(val))
            for _ in range(3):
                dev_denoise(ctx, x, args.wavelet, args.levels, args.boundary, -1.0, True, out=y)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(args.reps):
                dev_denoise(ctx, x, args.wavelet, args.levels, args.boundary, -1.0, True, out=y)
            e1.record(stream)
            torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / args.reps)
            cur = y.cpu().numpy().view(np.uint64).copy()
            if ref is None:
                ref = cur
            assert np.array_equal(ref, cur), "variant %s differs" % name
    for name, t in res.items():
        t = np.array(t)
        print("%-10s med %.4f ms  min %.4f ms  %.2f Gsamples/s  %.0f GB/s of 16 B/sample" % (
            name, np.median(t), t.min(), args.rows * args.n / np.median(t) / 1e6, 16.0 * args.rows * args.n / np.median(t) / 1e6))


if __name__ == "__main__":
    main()
