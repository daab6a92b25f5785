#!/usr/bin/env python3
"""Follow-up to exp_place.py: sub-page offsets of the coefficient block inside one arena, then separately
allocated blocks behind pads of varying size (physical placement)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vectorwave_amd import _native as N
from util import dev_forward, dev_inverse, dev_signals
ctx = N.Context(0)
rows, n, j = 65536, 16384, 6
MiB = 1 << 20
xe, ce = rows * n, rows * (j + 1) * n
x0 = dev_signals(rows, n, seed=1)
stream = torch.cuda.current_stream()
def timed(fn, k=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(k): fn()
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
def run(tag, x, c, xr):
    ff = timed(lambda: dev_forward(ctx, x, "db4", j, 0, N.VW_FLAG_FMA, out=c))
    fe = timed(lambda: dev_forward(ctx, x, "db4", j, 0, 0, out=c))
    ie = timed(lambda: dev_inverse(ctx, c, "db4", 0, 0, out=xr))
    fi = timed(lambda: dev_inverse(ctx, c, "db4", 0, N.VW_FLAG_FMA, out=xr))
    print("%-44s | fwd fma %.2f exact %.2f | inv exact %.2f fma %.2f" % (tag, ff, fe, ie, fi), flush=True)
part = os.environ.get("EXP_PART", "12")
xr0 = torch.empty((rows, n), dtype=torch.float64, device="cuda")
if "1" in part:
    arena = torch.empty(ce + 64 * MiB // 8, dtype=torch.float64, device="cuda")
    for off in (0, 64, 128, 192, 256, 512, 1024, 2048, 4096, 4160, 8320, 12480, 24 * MiB + 4160, 48 * MiB + 8320):
        c = arena[off // 8: off // 8 + ce].view(rows, j + 1, n)
        run("arena(c only) c+%d B" % off, x0, c, xr0)
    del arena, c
    torch.cuda.empty_cache()
if "2" in part:
    for pad_mib in (0, 2, 4, 8, 16, 24, 32, 40, 48, 56, 64, 24, 0):
        pad = torch.empty(max(1, pad_mib * MiB // 8), dtype=torch.float64, device="cuda")
        c = torch.empty((rows, j + 1, n), dtype=torch.float64, device="cuda")
        run("separate c behind a %d MiB pad, c=0x%x" % (pad_mib, c.data_ptr()), x0, c, xr0)
        del c, pad
        torch.cuda.empty_cache()
