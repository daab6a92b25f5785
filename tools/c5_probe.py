"""Timing probe of the single-launch row denoise: derived against fixed threshold, soft against hard, per filter."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from util import dev_signals, dev_denoise
from vectorwave_amd import _native as N
ctx = N.Context(0)
x = dev_signals(8192, 480, seed=6, noise=0.3); y = torch.empty_like(x)
def timed(f, k=200):
    for _ in range(10): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / k
for wn in ("db4", "haar", "db4"):
    for thr in (-1.0, 0.3):
        for soft in (True, False):
            print(wn, thr, soft, round(timed(lambda: dev_denoise(ctx, x, wn, 4, 0, thr, soft, out=y)), 4), flush=True)
