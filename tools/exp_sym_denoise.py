#!/usr/bin/env python3
"""Times the long-row WaveletDenoiser (UNIVERSAL, soft) and the SWT adapter's denoise per boundary mode.
usage: python tools/exp_sym_denoise.py [--rows 16384] [--n 16384]"""
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
    args = ap.parse_args()
    import torch
    from vectorwave_amd import _native as N
    from util import dev_denoise, dev_signals, dev_wavelet_denoise
    ctx = N.Context(0)
    stream = torch.cuda.current_stream()
    x = dev_signals(args.rows, args.n, seed=6, noise=0.3)
    y = torch.empty_like(x)
    for name, fn in (("WaveletDenoiser", lambda m: dev_wavelet_denoise(ctx, x, "db4", args.levels, m, 0, True, out=y)),
                     ("SWT adapter", lambda m: dev_denoise(ctx, x, "db4", args.levels, m, -1.0, True, out=y))):
        for mode, mname in ((0, "PERIODIC"), (1, "ZERO_PADDING"), (2, "SYMMETRIC")):
            fn(mode)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(4):
                fn(mode)
            e1.record(stream)
            torch.cuda.synchronize()
            print("%-16s %-13s %.3f ms   %s" % (name, mname, e0.elapsed_time(e1) / 4, ctx.last_plan()), flush=True)


if __name__ == "__main__":
    main()
