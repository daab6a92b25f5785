#!/usr/bin/env python3
"""Times every BASELINE configuration on one GPU through the device-pointer C ABI (inputs resident in
HBM, HIP events on the launch stream) and prints one JSON object.  Not the headline bench (bench.py is);
this is the per-configuration table of the design notes.

usage: python tools/bench_configs.py [--steps K] [--warmup W] [--out file.json]
"""
import argparse
import ctypes as C
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0
FP64_LANE_OPS = 256 * 4 * 16 * 2.4e9  # 39.3e12 fp64 lane-operations per second


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--out", default="")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import torch
    from vectorwave_amd import _native as N
    from util import bind_dev_forward, bind_dev_inverse, dev_denoise, dev_signals, dev_wavelet_denoise
    from vectorwave_amd.wavelets import get_wavelet

    ctx = N.Context(0)
    stream = torch.cuda.current_stream()

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.steps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.steps

    # (name, wavelet, rows, n, J, boundary, flags)
    transforms = [
        ("C1 db4 32x1000 single level", "db4", 32, 1000, 1, 0, N.VW_FLAG_SINGLE_LEVEL),
        ("C2 haar 4096x65536 J=6", "haar", 4096, 65536, 6, 0, 0),
        ("C2 db4 4096x65536 J=6", "db4", 4096, 65536, 6, 0, 0),
        ("C2 db4 4096x65536 J=6 ZERO_PADDING", "db4", 4096, 65536, 6, 1, 0),
        ("C2 db4 4096x65536 J=6 SYMMETRIC", "db4", 4096, 65536, 6, 2, 0),
        ("C3 sym8 64x777777 J=8 PERIODIC", "sym8", 64, 777777, 8, 0, 0),
        ("C3 sym8 64x777777 J=8 ZERO_PADDING", "sym8", 64, 777777, 8, 1, 0),
        ("C3 coif3 64x777777 J=8 PERIODIC", "coif3", 64, 777777, 8, 0, 0),
        ("C3 coif3 64x777777 J=8 ZERO_PADDING", "coif3", 64, 777777, 8, 1, 0),
        ("C3 db10 64x777777 J=8 PERIODIC", "db10", 64, 777777, 8, 0, 0),
        ("C3 db10 64x777777 J=8 ZERO_PADDING", "db10", 64, 777777, 8, 1, 0),
        ("C4 db4 65536x16384 J=6", "db4", 65536, 16384, 6, 0, 0),
    ]
    res = {}
    for name, wn, rows, n, j, mode, flags in transforms:
        if args.only and args.only not in name:
            continue
        x = dev_signals(rows, n, seed=1)
        c = torch.empty((rows, j + 1, n), dtype=torch.float64, device=x.device)
        xr = torch.empty_like(x)
        # (arguments marshalled once: at 32 x 1000 the Python side of a call otherwise costs more than the launch)
        f_ms = timed(bind_dev_forward(ctx, x, wn, j, mode, flags, c))
        i_ms = timed(bind_dev_inverse(ctx, c, wn, mode, flags, xr))
        samples = rows * n
        bytes_dir = 8.0 * (j + 2) * samples
        w_ = get_wavelet(wn)
        taps_total = len(w_.dec_lo) + len(w_.dec_hi)
        res[name] = {
            "forward_ms": round(f_ms, 4), "inverse_ms": round(i_ms, 4),
            "fwd_inv_Gsamples_per_s": round(samples / ((f_ms + i_ms) * 1e-3) / 1e9, 3),
            "forward_GBps": round(bytes_dir / (f_ms * 1e-3) / 1e9, 1),
            "inverse_GBps": round(bytes_dir / (i_ms * 1e-3) / 1e9, 1),
            "pct_of_hbm_roofline": round(100.0 * 2 * bytes_dir / ((f_ms + i_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 2),
            # the other bound: one multiply and one add per tap and filter in the reference's arithmetic, both directions,
            # against 256 CUs x 4 SIMDs x 16 fp64 lanes at 2.4 GHz (what decides the long-filter configurations)
            "fp64_lane_ops_per_sample": 2 * 2 * taps_total * j,
            "pct_of_fp64_issue": round(100.0 * 2 * 2 * taps_total * j * samples / ((f_ms + i_ms) * 1e-3) / FP64_LANE_OPS, 2),
            "max_reconstruction_error": float((xr - x).abs().max()) if mode == 0 else None,
        }
        print(name, res[name], flush=True)
        del x, c, xr
        torch.cuda.empty_cache()

    if not args.only or "C5" in args.only:
        rows, n, j = 8192, 480, 4
        x = dev_signals(rows, n, seed=5, noise=0.5)
        y = torch.empty_like(x)
        for wn in ("db4", "haar"):
            d_ms = timed(lambda: dev_denoise(ctx, x, wn, j, 0, -1.0, True, out=y))
            name = "C5 %s denoise 8192x480 J=4 universal soft" % wn
            res[name] = {
                "denoise_ms": round(d_ms, 4),
                "Gsamples_per_s": round(rows * n / (d_ms * 1e-3) / 1e9, 3),
                "blocks_per_s": round(rows / (d_ms * 1e-3)),
                # algorithmic traffic of a denoise that keeps coefficients on chip: read x, write y
                "algorithmic_GBps_16B_per_sample": round(16.0 * rows * n / (d_ms * 1e-3) / 1e9, 1),
            }
            print(name, res[name], flush=True)
    # WaveletDenoiser (the first "next" row): multi-level denoise with level-scaled thresholds
    if not args.only or "N1" in args.only:
        names = {0: "UNIVERSAL", 1: "SURE", 2: "MINIMAX", 3: "BAYES"}
        for rows, n, j, methods in ((16384, 16384, 6, (0, 2, 3)), (8192, 480, 4, (0, 1, 2, 3))):
            x = dev_signals(rows, n, seed=6, noise=0.3)
            y = torch.empty_like(x)
            for m in methods:
                d_ms = timed(lambda: dev_wavelet_denoise(ctx, x, "db4", j, 0, m, True, out=y))
                name = "N1 db4 WaveletDenoiser %dx%d J=%d %s soft" % (rows, n, j, names[m])
                res[name] = {"denoise_ms": round(d_ms, 4), "Gsamples_per_s": round(rows * n / (d_ms * 1e-3) / 1e9, 3)}
                print(name, res[name], flush=True)
            del x, y
            torch.cuda.empty_cache()
    # MODWTStreamingDenoiser: 8192 lock-step streams, one 480-sample block each per call, ring of 1024 per stream
    if not args.only or "N2" in args.only:
        import numpy as np
        from vectorwave_amd.wavelets import get_wavelet
        rows, n, window = 8192, 480, 1024
        w = get_wavelet("db4")
        dp = C.POINTER(C.c_double)
        taps = [np.ascontiguousarray(t, dtype=np.float64) for t in (w.dec_lo, w.dec_hi, w.rec_lo, w.rec_hi)]
        xs = [dev_signals(rows, n, seed=7 + q, noise=0.3) for q in range(8)]  # a new block per call, as a stream delivers
        x = xs[0]
        y = torch.empty_like(x)
        sp = C.c_void_p(stream.cuda_stream)
        calls = [0]
        for est, mult, label in ((0, 1.0, "MAD"), (0, 1.5, "MAD x1.5"), (1, 1.5, "STD x1.5"), (2, 1.0, "FIXED")):
            state = C.c_void_p()
            assert ctx.lib.vw_stream_state_create(ctx.handle, rows, window, C.byref(state)) == 0

            def run():
                calls[0] += 1
                st = ctx.lib.vw_streaming_denoise_batch_dev(
                    ctx.handle, state, C.c_void_p(xs[calls[0] % 8].data_ptr()), rows, n, taps[0].ctypes.data_as(dp), taps[0].size,
                    taps[1].ctypes.data_as(dp), taps[1].size, taps[2].ctypes.data_as(dp), taps[2].size,
                    taps[3].ctypes.data_as(dp), taps[3].size, 0, 0, 0, 1, mult, est, -1.0, -1.0, C.c_void_p(y.data_ptr()),
                    None, None, 0, sp)
                assert st == 0, ctx.last_error()
            ms = timed(run)
            torch.cuda.synchronize()
            ctx.lib.vw_stream_state_destroy(ctx.handle, state)
            name = "N2 db4 MODWTStreamingDenoiser %dx%d ring %d %s" % (rows, n, window, label)
            res[name] = {"block_ms": round(ms, 4), "Gsamples_per_s": round(rows * n / (ms * 1e-3) / 1e9, 3),
                         "blocks_per_s": round(rows / (ms * 1e-3))}
            print(name, res[name], flush=True)
    # BatchSIMDMODWT.batchMODWTSoA: single-level transform of a sample-major batch (24 B/sample)
    if not args.only or "N3" in args.only:
        from vectorwave_amd.wavelets import get_wavelet
        rows, n = 65536, 4096
        soa = torch.randn((n, rows), dtype=torch.float64, device="cuda")
        a = torch.empty_like(soa)
        d = torch.empty_like(soa)
        sp = C.c_void_p(stream.cuda_stream)
        dp = C.POINTER(C.c_double)
        for wn in ("haar", "db2", "db4"):
            w = get_wavelet(wn)
            import numpy as np
            h = np.ascontiguousarray(w.dec_lo, dtype=np.float64)
            g = np.ascontiguousarray(w.dec_hi, dtype=np.float64)

            def run():
                st = ctx.lib.vw_batch_modwt_soa_dev(ctx.handle, C.c_void_p(soa.data_ptr()), rows, n, h.ctypes.data_as(dp), h.size,
                                                    g.ctypes.data_as(dp), g.size, 1 if wn == "haar" else 0,
                                                    C.c_void_p(a.data_ptr()), C.c_void_p(d.data_ptr()), sp)
                assert st == 0, ctx.last_error()
            ms = timed(run)
            name = "N3 %s batchMODWTSoA %dx%d" % (wn, rows, n)
            res[name] = {"forward_ms": round(ms, 4), "Gsamples_per_s": round(rows * n / (ms * 1e-3) / 1e9, 3),
                         "algorithmic_GBps_24B_per_sample": round(24.0 * rows * n / (ms * 1e-3) / 1e9, 1)}
            print(name, res[name], flush=True)
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
