#!/usr/bin/env python3
"""Is the placement sensitivity of the forward tied to the power-of-two row length?  Times the forward (FMA and exact)
and the inverse for rows of 16384 samples and of slightly longer rows, per placement state (coefficient block allocated
behind pads of different sizes); times are normalised to 16384 samples per row."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vectorwave_amd import _native as N
from util import dev_forward, dev_inverse, dev_signals
ctx = N.Context(0)
rows, j = 65536, 6
MiB = 1 << 20
stream = torch.cuda.current_stream()
def timed(fn, k=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(k): fn()
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
for n in (16384, 16384 + 64, 16384 + 512, 16384 + 2048, 16384 - 2048):
    x0 = dev_signals(rows, n, seed=1)
    xr0 = torch.empty((rows, n), dtype=torch.float64, device="cuda")
    for pad_mib in (0, 4, 8, 16):
        pad = torch.empty(max(1, pad_mib * MiB // 8), dtype=torch.float64, device="cuda")
        c = torch.empty((rows, j + 1, n), dtype=torch.float64, device="cuda")
        ff = timed(lambda: dev_forward(ctx, x0, "db4", j, 0, N.VW_FLAG_FMA, out=c))
        fe = timed(lambda: dev_forward(ctx, x0, "db4", j, 0, 0, out=c))
        ie = timed(lambda: dev_inverse(ctx, c, "db4", 0, 0, out=xr0))
        fi = timed(lambda: dev_inverse(ctx, c, "db4", 0, N.VW_FLAG_FMA, out=xr0))
        k = 16384.0 / n
        print("n %6d pad %3d MiB | per 16384 samples: fwd fma %.2f exact %.2f | inv exact %.2f fma %.2f | plan %s" % (
            n, pad_mib, ff * k, fe * k, ie * k, fi * k, ctx.last_plan()), flush=True)
        del c, pad
        torch.cuda.empty_cache()
    del x0, xr0
    torch.cuda.empty_cache()
