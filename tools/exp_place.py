#!/usr/bin/env python3
"""Which address relation makes the headline kernels fast or slow?  One arena, allocated once (so the
virtual-to-physical mapping is fixed), holds x, the coefficient block and x'; the three are placed at
varying offsets inside it and the forward (FMA mode = memory-bound, and exact) and the inverse are timed."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vectorwave_amd import _native as N
from util import dev_forward, dev_inverse, dev_signals
ctx = N.Context(0)
rows, n, j = 65536, 16384, 6
MiB = 1 << 20
slack = 160 * MiB
xe, ce = rows * n, rows * (j + 1) * n
arena = torch.empty((2 * xe + ce) + 3 * slack // 8, dtype=torch.float64, device="cuda")
x0 = dev_signals(rows, n, seed=1)
stream = torch.cuda.current_stream()
def timed(fn, k=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(k): fn()
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
def place(xo, co, ro):
    """byte offsets of x, c and x' inside their slots of the arena"""
    bx = xo // 8
    bc = xe + slack // 8 + co // 8
    br = xe + slack // 8 + ce + slack // 8 + ro // 8
    return arena[bx: bx + xe].view(rows, n), arena[bc: bc + ce].view(rows, j + 1, n), arena[br: br + xe].view(rows, n)
print("arena base 0x%x" % arena.data_ptr(), flush=True)
cases = []
for m in (0, 1, 2, 4, 8, 12, 16, 20, 24, 28, 32, 48, 64, 96, 128): cases.append((0, m * MiB, 0))
for k in (4, 16, 64, 128, 256, 512): cases.append((0, k << 10, 0))
for m in (1, 2, 4, 8, 16, 24, 32): cases.append((m * MiB, 0, 0))
for m in (1, 2, 4, 8, 16, 24, 32): cases.append((0, 0, m * MiB))
for m in (8, 16, 24): cases.append((m * MiB, m * MiB, m * MiB))
only = os.environ.get("EXP_ONLY")
for xo, co, ro in cases:
    x, c, xr = place(xo, co, ro)
    x.copy_(x0)
    ff = timed(lambda: dev_forward(ctx, x, "db4", j, 0, N.VW_FLAG_FMA, out=c))
    fe = timed(lambda: dev_forward(ctx, x, "db4", j, 0, 0, out=c))
    ie = timed(lambda: dev_inverse(ctx, c, "db4", 0, 0, out=xr))
    fi = timed(lambda: dev_inverse(ctx, c, "db4", 0, N.VW_FLAG_FMA, out=xr))
    print("x+%8.3f MiB c+%8.3f MiB xr+%8.3f MiB | fwd fma %.2f exact %.2f | inv exact %.2f fma %.2f" % (
        xo / MiB, co / MiB, ro / MiB, ff, fe, ie, fi), flush=True)
