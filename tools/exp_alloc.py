#!/usr/bin/env python3
"""Does the forward kernel's time depend on where its buffers land?  Re-allocates the coefficient block
at shifted addresses inside one process and times the FMA-mode forward (the memory-bound one)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vectorwave_amd import _native as N
from util import dev_forward, dev_inverse, dev_signals
ctx = N.Context(0)
rows, n, j = 65536, 16384, 6
x = dev_signals(rows, n, seed=1)
stream = torch.cuda.current_stream()
def timed(fn, k=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(k): fn()
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
keep = []
for trial in range(8):
    pad_elems = trial * (3 << 20) + trial * 520  # shifts the base by a non-power-of-two amount
    if os.environ.get("EXP_ARENA"):
        arena = torch.empty(rows * (j + 1) * n + pad_elems + 64, dtype=torch.float64, device="cuda")
        c = arena[pad_elems & ~1: (pad_elems & ~1) + rows * (j + 1) * n].view(rows, j + 1, n)
    else:
        pad = torch.empty(max(1, pad_elems), dtype=torch.float64, device="cuda")
        keep.append(pad)
        c = torch.empty((rows, j + 1, n), dtype=torch.float64, device="cuda")
    f = timed(lambda: dev_forward(ctx, x, "db4", j, 0, N.VW_FLAG_FMA, out=c))
    fe = timed(lambda: dev_forward(ctx, x, "db4", j, 0, 0, out=c))
    fz = 0.0 if os.environ.get("EXP_NOTORCH") else timed(lambda: c.zero_())
    c2 = c.view(-1)[: rows * n].view(rows, n)
    fc = 1.0 if os.environ.get("EXP_NOTORCH") else timed(lambda: c2.copy_(x))
    if not os.environ.get("EXP_NOTORCH"): print("   torch: zero_ of the block %.3f ms (%.0f GB/s), copy_ x->block plane %.3f ms (%.0f GB/s)" % (fz, c.numel() * 8 / fz / 1e6, fc, 2 * x.numel() * 8 / fc / 1e6))
    print("trial %d base 0x%x (mod 2MiB = %d KiB)  fwd fma %.3f ms  exact %.3f ms" % (trial, c.data_ptr(), (c.data_ptr() % (2 << 20)) >> 10, f, fe), flush=True)
    del c
    if os.environ.get("EXP_ARENA"): del arena
    torch.cuda.empty_cache()
