#!/usr/bin/env python3
"""Forward (FMA mode: memory-bound) on coefficient blocks allocated behind pads of different sizes; prints the
time of each group of launches so that per-dispatch counters of a rocprofv3 --pmc pass over this script can be
attributed to the fast and the slow placements."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vectorwave_amd import _native as N
from util import dev_forward, dev_inverse, dev_signals
ctx = N.Context(0)
rows, n, j = 65536, 16384, 6
MiB = 1 << 20
x0 = dev_signals(rows, n, seed=1)
xr0 = torch.empty((rows, n), dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream()
def timed(fn, k=3):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(k): fn()
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
for pad_mib in (0, 4, 8, 32, 16, 64):
    pad = torch.empty(max(1, pad_mib * MiB // 8), dtype=torch.float64, device="cuda")
    c = torch.empty((rows, j + 1, n), dtype=torch.float64, device="cuda")
    ff = timed(lambda: dev_forward(ctx, x0, "db4", j, 0, N.VW_FLAG_FMA, out=c))
    fi = timed(lambda: dev_inverse(ctx, c, "db4", 0, N.VW_FLAG_FMA, out=xr0))
    print("pad %3d MiB: 3 x fwd fma %.2f ms, 3 x inv fma %.2f ms" % (pad_mib, ff, fi), flush=True)
    del c, pad
    torch.cuda.empty_cache()
