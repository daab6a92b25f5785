#!/usr/bin/env python3
"""Forward time vs the distance between the input plane and the coefficient block inside one arena."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vectorwave_amd import _native as N
from util import dev_forward, dev_inverse, dev_signals
ctx = N.Context(0)
rows, n, j = 65536, 16384, 6
xs = dev_signals(rows, n, seed=1)
stream = torch.cuda.current_stream()
def timed(fn, k=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(k): fn()
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
maxgap = (256 << 20) // 8
arena = torch.empty(rows * n * (j + 2) + maxgap + 64, dtype=torch.float64, device="cuda")
x = arena[: rows * n].view(rows, n)
x.copy_(xs)
del xs
gaps_kib = [int(g) for g in os.environ.get("GAPS", "0 4 16 64 128 256 512 1024 2048 4096 8192 16384 32768 65536 131072 896 1792 3584 100 1000 10000").split()]
print("arena base 0x%x" % arena.data_ptr())
for g in gaps_kib:
    off = rows * n + g * 128
    c = arena[off: off + rows * (j + 1) * n].view(rows, j + 1, n)
    f = timed(lambda: dev_forward(ctx, x, "db4", j, 0, N.VW_FLAG_FMA, out=c))
    fe = timed(lambda: dev_forward(ctx, x, "db4", j, 0, 0, out=c))
    print("gap %7d KiB  fwd fma %.3f ms  exact %.3f ms" % (g, f, fe), flush=True)
