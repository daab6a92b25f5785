#!/usr/bin/env python3
"""Forward time per placement state (coefficient block allocated behind pads of different sizes) and per plan
variant (run length, tile shape): is there a variant whose time does not depend on the placement?"""
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
stream = torch.cuda.current_stream()
def timed(fn, k=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(k): fn()
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
variants = [("default", {}), ("c4", {"fwd_chunks": 4, "fwd_rot": 0}), ("c4r4", {"fwd_chunks": 4, "fwd_rot": 4}), ("c4r2", {"fwd_chunks": 4, "fwd_rot": 2}),
            ("c6r6", {"fwd_chunks": 6, "fwd_rot": 6}), ("c8", {"fwd_chunks": 8, "fwd_rot": 0}), ("c8r8", {"fwd_chunks": 8, "fwd_rot": 8}), ("c8r4", {"fwd_chunks": 8, "fwd_rot": 4}),
            ("c16", {"fwd_chunks": 16, "fwd_rot": 0}), ("c16r16", {"fwd_chunks": 16, "fwd_rot": 16}), ("c16r8", {"fwd_chunks": 16, "fwd_rot": 8}), ("c16r4", {"fwd_chunks": 16, "fwd_rot": 4})]
if os.environ.get("EXP_VARIANTS"):
    variants = [(v, dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in v.split(",") if kv)) for v in os.environ["EXP_VARIANTS"].split(";")]
flag = N.VW_FLAG_FMA if os.environ.get("EXP_FMA", "1") == "1" else 0
print("pad   " + " ".join("%8s" % v[0] for v in variants))
for pad_mib in (0, 4, 8, 16, 32, 64):
    pad = torch.empty(max(1, pad_mib * MiB // 8), dtype=torch.float64, device="cuda")
    c = torch.empty((rows, j + 1, n), dtype=torch.float64, device="cuda")
    res = []
    for name, opts in variants:
        ctx.set_plan_option("reset", 0)
        for k, v in opts.items(): ctx.set_plan_option(k, v)
        res.append(timed(lambda: dev_forward(ctx, x0, "db4", j, 0, flag, out=c)))
    print("%3d   " % pad_mib + " ".join("%8.2f" % r for r in res), flush=True)
    del c, pad
    torch.cuda.empty_cache()
