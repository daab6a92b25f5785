#!/usr/bin/env python3
"""Headline benchmark: batched DB4 J=6 fp64 MODWT forward+inverse (PERIODIC).

One "step" = one forward (x[B][N] -> coeffs[B][J+1][N]) plus one inverse
(coeffs -> x'[B][N]) over a synthetic batch that is already resident in HBM.
Default workload at 1 GPU is BASELINE.json's metric configuration: 65536 signals
x 16384 samples.  With --gpus N every rank (one process per GPU, launched by
torch.distributed.run) transforms its own batch of that size; signals are
independent, so there is no data-path collective -- torch.distributed is used
only for the barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line (rank 0).  Fields beyond the driver contract:
  roofline     -- the slower of the two kernels of a step, algorithmic bytes
                  (8*(J+2) per sample per direction) / its mean launch duration,
                  measured with device events on the stream the kernels run on
  cpu_baseline -- the CPU oracle (a C restatement of the reference's Java
                  algorithm, OpenMP over signals) timed on a bounded sample of
                  the same workload on this host's cores.  Reported, not a target.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=65536, help="signals per GPU")
    ap.add_argument("--n", type=int, default=16384, help="samples per signal")
    ap.add_argument("--levels", type=int, default=6)
    ap.add_argument("--wavelet", default="db4")
    ap.add_argument("--boundary", type=int, default=0)
    ap.add_argument("--arith", choices=["exact", "fma"], default="exact",
                    help="exact: the reference's arithmetic, bit parity (default); fma: VW_FLAG_FMA")
    ap.add_argument("--no-alt-mode", action="store_true", help="skip the secondary run in the other arithmetic mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the CPU-baseline sample (0 = auto)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from vectorwave_amd import _native as N
    from vectorwave_amd.wavelets import get_wavelet

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MODWT engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under a launcher (RANK/WORLD_SIZE set) the process group is always created, also for one rank, so
    # the single-GPU box exercises the same barrier / MAX-over-ranks code as the 8-GPU node
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)

    w = get_wavelet(args.wavelet)
    J, n = args.levels, args.n
    rows = args.rows
    # fit the batch to free HBM: x + coeffs + x' = (J+3) planes of rows*n doubles
    free_b, _total = torch.cuda.mem_get_info(dev)
    need = rows * n * 8 * (J + 3)
    while need > 0.92 * free_b and rows > 1:
        rows //= 2
        need = rows * n * 8 * (J + 3)

    ctx = N.Context(local_rank)
    lib = ctx.lib
    dp = C.POINTER(C.c_double)
    h = np.ascontiguousarray(w.dec_lo, dtype=np.float64)
    g = np.ascontiguousarray(w.dec_hi, dtype=np.float64)
    hr = np.ascontiguousarray(w.rec_lo, dtype=np.float64)
    gr = np.ascontiguousarray(w.rec_hi, dtype=np.float64)

    # synthetic batch generated on the device: sin(2 pi t/32) + 0.5 sin(2 pi t/64) + 0.1 N(0,1)
    x = torch.empty((rows, n), dtype=torch.float64, device=dev)
    t = torch.arange(n, dtype=torch.float64, device=dev)
    base = torch.sin(2 * math.pi * t / 32.0) + 0.5 * torch.sin(2 * math.pi * t / 64.0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(42 + rank)
    chunk = max(1, (256 << 20) // (n * 8))
    for r0 in range(0, rows, chunk):
        r1 = min(rows, r0 + chunk)
        x[r0:r1] = base[None, :] + 0.1 * torch.randn((r1 - r0, n), dtype=torch.float64, device=dev, generator=gen)
    coeffs = torch.empty((rows, J + 1, n), dtype=torch.float64, device=dev)
    xr = torch.empty((rows, n), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)

    mode_flags = {"exact": 0, "fma": N.VW_FLAG_FMA}
    flags = [mode_flags[args.arith]]

    def fwd():
        st = lib.vw_modwt_forward_batch_dev(ctx.handle, C.c_void_p(x.data_ptr()), rows, n, J, h.ctypes.data_as(dp),
                                            h.size, g.ctypes.data_as(dp), g.size, args.boundary, flags[0],
                                            C.c_void_p(coeffs.data_ptr()), None, sp)
        if st != 0:
            raise RuntimeError("forward failed: %d %s" % (st, ctx.last_error()))

    def inv():
        st = lib.vw_modwt_inverse_batch_dev(ctx.handle, C.c_void_p(coeffs.data_ptr()), rows, n, J,
                                            hr.ctypes.data_as(dp), hr.size, gr.ctypes.data_as(dp), gr.size,
                                            args.boundary, flags[0], C.c_void_p(xr.data_ptr()), None, sp)
        if st != 0:
            raise RuntimeError("inverse failed: %d %s" % (st, ctx.last_error()))

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        fwd()
        inv()
    barrier()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    t_start = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        fwd()
        ev[k][1].record(stream)
        inv()
        ev[k][2].record(stream)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t_start
    barrier()
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    fwd_ms = float(np.mean([ev[k][0].elapsed_time(ev[k][1]) for k in range(args.steps)]))
    inv_ms = float(np.mean([ev[k][1].elapsed_time(ev[k][2]) for k in range(args.steps)]))

    # ---- sanity: perfect reconstruction at full size, parity with the oracle on a few rows ------------
    err = float((xr[: min(rows, 4096)] - x[: min(rows, 4096)]).abs().max().item())
    verified = None
    max_dev = None
    cpu = None
    if rank == 0:
        from oracle import oracle as o
        xs = x[:2].cpu().numpy()
        st, want = o.decompose_batch(xs, w, J, args.boundary)
        got = coeffs[:2].cpu().numpy()
        wantx = o.reconstruct_batch(want, w, args.boundary)
        gotx = xr[:2].cpu().numpy()
        verified = bool(st == 0 and np.array_equal(got.view(np.uint64), want.view(np.uint64))
                        and np.array_equal(gotx.view(np.uint64), wantx.view(np.uint64)))
        # north-star tolerance: |gpu - ref| <= 1e-12 * max(|ref|, max|x|)
        scale = float(np.abs(xs).max())
        max_dev = float(max((np.abs(got - want) / np.maximum(np.abs(want), scale)).max(),
                            (np.abs(gotx - wantx) / np.maximum(np.abs(wantx), scale)).max()))
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(o, w, J, n, args)

    # ---- secondary run in the other arithmetic mode (reported beside the headline, never as `value`) ----
    alt = None
    if not args.no_alt_mode:
        other = "fma" if args.arith == "exact" else "exact"
        flags[0] = mode_flags[other]
        for _ in range(max(1, args.warmup)):
            fwd()
            inv()
        barrier()
        ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        nrep = max(3, args.steps // 2)
        ev2[0].record(stream)
        for _ in range(nrep):
            fwd()
        ev2[1].record(stream)
        for _ in range(nrep):
            inv()
        ev2[2].record(stream)
        torch.cuda.synchronize(dev)
        f2, i2 = ev2[0].elapsed_time(ev2[1]) / nrep, ev2[1].elapsed_time(ev2[2]) / nrep
        alt = {"arith": other, "forward_ms": round(f2, 4), "inverse_ms": round(i2, 4),
               "Gsamples_per_s_per_gpu": round(rows * n / ((f2 + i2) * 1e-3) / 1e9, 4),
               "pct_of_hbm_roofline": round(100.0 * rows * n / ((f2 + i2) * 1e-3) / 1e9 * 16 * (J + 2) / HBM_PEAK_GBS, 2)}
        if rank == 0:
            got2 = coeffs[:2].cpu().numpy()
            gotx2 = xr[:2].cpu().numpy()
            alt["max_rel_dev_vs_oracle"] = float(max((np.abs(got2 - want) / np.maximum(np.abs(want), scale)).max(),
                                                     (np.abs(gotx2 - wantx) / np.maximum(np.abs(wantx), scale)).max()))
        flags[0] = mode_flags[args.arith]

    samples = rows * n * args.steps * world
    value = samples / elapsed / 1e9
    bytes_per_dir = 8.0 * (J + 2) * rows * n
    dom_name, dom_ms = ("k_inv_fused", inv_ms) if inv_ms >= fwd_ms else ("k_fwd_stream", fwd_ms)
    achieved = bytes_per_dir / (dom_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = "%s:%dx%d:J%d:%s" % (dom_name, rows, n, J, args.wavelet)
            traffic = tj.get(key)
        except Exception:
            traffic = None
    out = {
        "metric": "MODWT fwd+inv Gsamples/s (whole node), DB4 J=6 fp64 batch; % of HBM roofline",
        "value": round(value, 4),
        "unit": "Gsamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "%s J=%d MODWT forward+inverse, %s, %d signals x %d samples per GPU, coefficients materialised"
                        % (args.wavelet.upper(), J, ["PERIODIC", "ZERO_PADDING", "SYMMETRIC"][args.boundary], rows, n),
            "signals_per_gpu": rows, "samples_per_signal": n, "levels": J, "wavelet": args.wavelet,
            "parallelism": "rows sharded over %d GPU(s), no collective" % world,
            "bytes_per_sample_fwd_plus_inv": 16 * (J + 2),
        },
        "pct_of_hbm_roofline": round(100.0 * value / world * 16 * (J + 2) / HBM_PEAK_GBS, 2),
        "forward_ms": round(fwd_ms, 4),
        "inverse_ms": round(inv_ms, 4),
        "arith": args.arith,
        "max_reconstruction_error": err,
        "bit_exact_vs_oracle": verified,
        "max_rel_dev_vs_oracle": max_dev,
        "alt_mode": alt,
        "roofline": {
            "bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": bytes_per_dir,
            "other_kernel": {"kernel": "k_fwd_stream" if dom_name == "k_inv_fused" else "k_inv_fused",
                             "achieved": round(bytes_per_dir / (min(fwd_ms, inv_ms) * 1e-3) / 1e9, 1)},
        },
        "cpu_baseline": cpu,
    }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(o, w, J, n, args):
    """Times the oracle's batch drivers (OpenMP over signals) on a bounded sample."""
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    rng = np.random.default_rng(7)
    t = np.arange(n)
    base = np.sin(2 * np.pi * t / 32.0) + 0.5 * np.sin(2 * np.pi * t / 64.0)

    def run(rows):
        xs = base[None, :] + 0.1 * rng.standard_normal((rows, n))
        t0 = time.perf_counter()
        st, c = o.decompose_batch(xs, w, J, args.boundary)
        o.reconstruct_batch(c, w, args.boundary)
        return time.perf_counter() - t0

    rows = args.cpu_rows
    if rows <= 0:
        probe = max(cores, 8)
        tp = run(probe)
        rows = int(max(probe, min(32768, probe * 15.0 / max(tp, 1e-6))))  # aim at ~15 s
    el = run(rows)
    return {
        "value": round(rows * n / el / 1e9, 5), "unit": "Gsamples/s", "cores": cores, "kind": "port",
        "sample": "%d signals x %d samples, %s J=%d fwd+inv, C restatement of the reference's scalar/vector-order "
                  "Java loops (dense zero-stuffed inverse), gcc -O3 -mavx2 -fopenmp, %.1f s"
                  % (rows, n, args.wavelet.upper(), J, el),
    }


if __name__ == "__main__":
    main()
