#!/usr/bin/env python3
"""Compare the wave-private inverse with the default plan on a small batch and show where they differ."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vectorwave_amd import _native as N
from vectorwave_amd.wavelets import get_wavelet
rows, n, J = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 16384, 6
nch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda", 0)
ctx = N.Context(0)
w = get_wavelet("db4")
dp = C.POINTER(C.c_double)
h, g, hr, gr = (np.ascontiguousarray(v, dtype=np.float64) for v in (w.dec_lo, w.dec_hi, w.rec_lo, w.rec_hi))
torch.manual_seed(1)
x = torch.randn((rows, n), dtype=torch.float64, device=dev)
coeffs = torch.empty((rows, J + 1, n), dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream(dev)
sp = C.c_void_p(stream.cuda_stream)
st = ctx.lib.vw_modwt_forward_batch_dev(ctx.handle, C.c_void_p(x.data_ptr()), rows, n, J, h.ctypes.data_as(dp), h.size, g.ctypes.data_as(dp), g.size, 0, 0, C.c_void_p(coeffs.data_ptr()), None, sp)
assert st == 0
def inv(opts):
    ctx.set_plan_option("reset", 0)
    for k, v in opts: ctx.set_plan_option(k, v)
    xr = torch.full((rows, n), float("nan"), dtype=torch.float64, device=dev)
    st = ctx.lib.vw_modwt_inverse_batch_dev(ctx.handle, C.c_void_p(coeffs.data_ptr()), rows, n, J, hr.ctypes.data_as(dp), hr.size, gr.ctypes.data_as(dp), gr.size, 0, 0, C.c_void_p(xr.data_ptr()), None, sp)
    assert st == 0, ctx.last_error()
    torch.cuda.synchronize()
    return xr.cpu().numpy(), ctx.last_plan() if hasattr(ctx, "last_plan") else ""
a, pa = inv([])
b, pb = inv([("inv_wave", nch)])
print("plans:", pa, "|", pb)
bad = a.view(np.uint64) != b.view(np.uint64)
print("mismatches:", bad.sum(), "of", bad.size, " nan in wave:", np.isnan(b).sum(), " max abs diff:", np.nanmax(np.abs(a - b)))
for r in range(min(rows, 2)):
    idx = np.nonzero(bad[r])[0]
    if idx.size == 0:
        print("row", r, "ok"); continue
    ch = idx // 512
    print("row", r, "bad count", idx.size, "chunks with errors:", np.unique(ch)[:40], "...")
    cnt = np.bincount(ch, minlength=n // 512)
    first = [int(idx[ch == c][0] % 512) if cnt[c] else -1 for c in range(n // 512)]
    print("  per-chunk bad counts:", cnt.tolist())
    print("  per-chunk first bad offset:", first)
    c0 = ch[0]
    inch = idx[ch == c0] % 512
    print("  in chunk", c0, "bad offsets (first 64):", inch[:64], "count", inch.size)
    print("  sample diffs:", [(int(i), float(a[r, i]), float(b[r, i])) for i in idx[:4]])
