#!/usr/bin/env python3
"""Headline benchmark: batched DB4 J=6 fp64 MODWT forward+inverse (PERIODIC).

One "step" = one forward (x[B][N] -> coeffs[B][J+1][N]) plus one inverse
(coeffs -> x'[B][N]) over a synthetic batch that is already resident in HBM.
The workload is BASELINE.json's metric configuration: 65536 signals x 16384
samples in all.  With --gpus N the 65536 rows are split into N contiguous row
blocks (vectorwave_amd.sharding.shard_rows: 8192 rows per GPU at N=8), one
process per GPU -- strong scaling.  Signals are independent (the reference's
batch loop carries no state between rows: modwt/MODWTTransform.java:564-567), so
there is no data-path collective; torch.distributed is used only for the
barrier and the max-over-ranks of the elapsed time.  `--scaling weak` keeps
the full batch on every GPU instead; the default run reports that figure too
(`other_scaling`) when more than one rank runs.

Launching: under a launcher (RANK/WORLD_SIZE in the environment, e.g.
torch.distributed.run) the process is one rank.  Without one, `--gpus N` with
N > 1 makes this process start N rank processes itself (before anything
touches a GPU), wait for them and exit with their status; the JSON line is
printed by rank 0.

Prints ONE JSON line (rank 0).  Fields beyond the driver contract:
  roofline     -- the slower of the two kernels of a step, algorithmic bytes
                  (8*(J+2) per sample per direction) / its mean launch duration,
                  measured with device events on the stream the kernels run on
  cpu_baseline -- CPU restatements of the reference's Java algorithm timed on a
                  bounded sample of the same workload on this host's cores: the
                  per-signal scalar path (OpenMP over signals) and, as
                  `cpu_baseline_soa`, the reference's SIMD batch scheme
                  (structure-of-arrays, lanes along the batch).  Reported, not targets; single-GPU
                  runs only.
"""
import argparse
import ctypes as C
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=65536,
                    help="signals of the whole job (strong scaling) or per GPU (weak scaling)")
    ap.add_argument("--n", type=int, default=16384, help="samples per signal")
    ap.add_argument("--levels", type=int, default=6)
    ap.add_argument("--wavelet", default="db4")
    ap.add_argument("--boundary", type=int, default=0)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: --rows split over the GPUs (default); weak: --rows on every GPU")
    ap.add_argument("--no-other-scaling", action="store_true", help="skip the secondary run in the other scaling mode")
    ap.add_argument("--arith", choices=["exact", "fma"], default="exact",
                    help="exact: the reference's arithmetic, bit parity (default); fma: VW_FLAG_FMA")
    ap.add_argument("--no-alt-mode", action="store_true", help="skip the secondary run in the other arithmetic mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the CPU-baseline sample (0 = auto)")
    ap.add_argument("--plan", action="append", default=[], metavar="NAME=VALUE",
                    help="vw_set_plan_option before the run (A/B of equivalent plans, e.g. --plan no_blk=1)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: ranks rendezvous over gloo, shard the rows, time a stub and print the line "
                         "(exercises the launcher, the sharding and the timing reduction on a CPU box)")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------
# launcher: one process per GPU
# ---------------------------------------------------------------------------------------------------

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """Start n rank processes of this script (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set), wait for all of them
    and return the exit status of the job: 0 only if every rank exited 0.  The first failing rank ends the
    others (they would otherwise wait in a collective for a peer that is gone).  The parent makes no GPU
    call."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "VW_BENCH_SPAWNED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    status = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                sys.stderr.write("bench.py: rank %d of %d exited with status %d; stopping the other ranks\n" % (r, n, rc))
                for o in alive:
                    procs[o].terminate()
        if alive:
            time.sleep(0.05)
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    return status


# ---------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------

def rank_rows(args, world, rank, scaling):
    from vectorwave_amd.sharding import shard_rows
    if scaling == "weak":
        return args.rows
    r0, r1 = shard_rows(args.rows, world, rank)
    return r1 - r0


def workload_text(args, J, rows_total, world, scaling):
    return ("%s J=%d MODWT forward+inverse, %s, %d signals x %d samples in all, %s over %d GPU(s), coefficients "
            "materialised" % (args.wavelet.upper(), J, ["PERIODIC", "ZERO_PADDING", "SYMMETRIC"][args.boundary],
                              rows_total, args.n,
                              "row blocks split" if scaling == "strong" else "the full batch on each", world))


def dry_run(args, world, rank):
    """The launcher / sharding / reduction path of a rank without a GPU (gloo)."""
    import torch
    import torch.distributed as dist
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == args.dry_run_fail_rank:
        raise SystemExit("bench.py: rank %d fails on request" % rank)
    rows = rank_rows(args, world, rank, args.scaling)
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (rank + 1))  # stands in for one forward + inverse
    elapsed = time.perf_counter() - t0
    total = torch.tensor([float(rows)], dtype=torch.float64)
    tmax = torch.tensor([elapsed], dtype=torch.float64)
    if use_dist:
        dist.barrier()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    rows_total = int(total.item())
    if rank == 0:
        print(json.dumps({
            "metric": "MODWT fwd+inv Gsamples/s (whole node), DB4 J=6 fp64 batch; % of HBM roofline",
            "value": rows_total * args.n * args.steps / float(tmax.item()) / 1e9, "unit": "Gsamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(tmax.item()) / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "dry_run": True,
            "config": {"workload": workload_text(args, args.levels, rows_total, world, args.scaling),
                       "signals_total": rows_total, "signals_rank0": rows, "samples_per_signal": args.n}}), flush=True)
    if use_dist:
        dist.destroy_process_group()


def main():
    args = parse()
    in_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if not in_launcher and args.gpus > 1:
        # be the launcher: nothing here may touch a GPU
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, world, rank)

    import numpy as np
    import torch
    import torch.distributed as dist

    from vectorwave_amd import _native as N
    from vectorwave_amd.wavelets import get_wavelet

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MODWT engine has no CPU path")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        raise SystemExit("bench.py: rank %d needs GPU %d but this node shows %d GPU(s); --gpus %d cannot run here"
                         % (rank, local_rank, ndev, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under a launcher the process group is always created, also for one rank, so the single-GPU box
    # exercises the same barrier / MAX-over-ranks code as the 8-GPU node
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)

    w = get_wavelet(args.wavelet)
    J, n = args.levels, args.n
    other = "weak" if args.scaling == "strong" else "strong"
    do_other = world > 1 and not args.no_other_scaling
    rows_main = rank_rows(args, world, rank, args.scaling)
    rows_other = rank_rows(args, world, rank, other) if do_other else 0
    rows_alloc = max(rows_main, rows_other, 1)
    # fit the batch to free HBM: x + coeffs + x' = (J+3) planes of rows*n doubles
    free_b, _total = torch.cuda.mem_get_info(dev)
    shrink = 1
    while rows_alloc * n * 8 * (J + 3) > 0.92 * free_b and rows_alloc > 1:
        rows_alloc //= 2
        shrink *= 2
    if shrink > 1:
        rows_main = max(1, rows_main // shrink) if rows_main > rows_alloc else rows_main
        rows_other = max(1, rows_other // shrink) if rows_other > rows_alloc else rows_other

    ctx = N.Context(local_rank)
    lib = ctx.lib
    for opt in args.plan:
        name, _, val = opt.partition("=")
        ctx.set_plan_option(name, int(val))
    dp = C.POINTER(C.c_double)
    h = np.ascontiguousarray(w.dec_lo, dtype=np.float64)
    g = np.ascontiguousarray(w.dec_hi, dtype=np.float64)
    hr = np.ascontiguousarray(w.rec_lo, dtype=np.float64)
    gr = np.ascontiguousarray(w.rec_hi, dtype=np.float64)

    # synthetic batch generated on the device: sin(2 pi t/32) + 0.5 sin(2 pi t/64) + 0.1 N(0,1)
    x = torch.empty((rows_alloc, n), dtype=torch.float64, device=dev)
    t = torch.arange(n, dtype=torch.float64, device=dev)
    base = torch.sin(2 * math.pi * t / 32.0) + 0.5 * torch.sin(2 * math.pi * t / 64.0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(42 + rank)
    chunk = max(1, (256 << 20) // (n * 8))
    for r0 in range(0, rows_alloc, chunk):
        r1 = min(rows_alloc, r0 + chunk)
        x[r0:r1] = base[None, :] + 0.1 * torch.randn((r1 - r0, n), dtype=torch.float64, device=dev, generator=gen)
    coeffs = torch.empty((rows_alloc, J + 1, n), dtype=torch.float64, device=dev)
    xr = torch.empty((rows_alloc, n), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)

    mode_flags = {"exact": 0, "fma": N.VW_FLAG_FMA}
    state = {"flags": mode_flags[args.arith], "rows": rows_main}

    def fwd():
        st = lib.vw_modwt_forward_batch_dev(ctx.handle, C.c_void_p(x.data_ptr()), state["rows"], n, J, h.ctypes.data_as(dp),
                                            h.size, g.ctypes.data_as(dp), g.size, args.boundary, state["flags"],
                                            C.c_void_p(coeffs.data_ptr()), None, sp)
        if st != 0:
            raise RuntimeError("forward failed: %d %s" % (st, ctx.last_error()))

    def inv():
        st = lib.vw_modwt_inverse_batch_dev(ctx.handle, C.c_void_p(coeffs.data_ptr()), state["rows"], n, J,
                                            hr.ctypes.data_as(dp), hr.size, gr.ctypes.data_as(dp), gr.size,
                                            args.boundary, state["flags"], C.c_void_p(xr.data_ptr()), None, sp)
        if st != 0:
            raise RuntimeError("inverse failed: %d %s" % (st, ctx.last_error()))

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed(steps, warmup):
        """`warmup` untimed steps, then exactly `steps` steps between barriers; returns (max-over-ranks seconds,
        mean forward ms, mean inverse ms, rows of all ranks)."""
        for _ in range(warmup):
            fwd()
            inv()
        barrier()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
        t_start = time.perf_counter()
        for k in range(steps):
            ev[k][0].record(stream)
            fwd()
            ev[k][1].record(stream)
            inv()
            ev[k][2].record(stream)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t_start
        barrier()
        tot = torch.tensor([float(state["rows"])], dtype=torch.float64, device=dev)
        if use_dist:
            tmax = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            el = float(tmax.item())
        f_ms = float(np.mean([ev[k][0].elapsed_time(ev[k][1]) for k in range(steps)]))
        i_ms = float(np.mean([ev[k][1].elapsed_time(ev[k][2]) for k in range(steps)]))
        return el, f_ms, i_ms, int(tot.item())

    elapsed, fwd_ms, inv_ms, rows_total = timed(args.steps, args.warmup)
    # the kernels this plan launched, as the library names them: "forward: k_fwd_stream[1-6]; inverse: k_inv_ring[1-6]"
    plan_txt = ctx.last_plan()
    rows = rows_main

    # ---- sanity: perfect reconstruction at full size, parity with the oracle on a few rows ------------
    err = float((xr[: min(rows, 4096)] - x[: min(rows, 4096)]).abs().max().item())
    verified = None
    max_dev = None
    cpu = None
    cpu_soa = None
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

    # ---- secondary run in the other arithmetic mode (reported beside the headline, never as `value`) ----
    alt = None
    if not args.no_alt_mode:
        other_arith = "fma" if args.arith == "exact" else "exact"
        state["flags"] = mode_flags[other_arith]
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
        alt = {"arith": other_arith, "forward_ms": round(f2, 4), "inverse_ms": round(i2, 4),
               "Gsamples_per_s_per_gpu": round(rows * n / ((f2 + i2) * 1e-3) / 1e9, 4),
               "pct_of_hbm_roofline": round(100.0 * rows * n / ((f2 + i2) * 1e-3) / 1e9 * 16 * (J + 2) / HBM_PEAK_GBS, 2)}
        if rank == 0:
            got2 = coeffs[:2].cpu().numpy()
            gotx2 = xr[:2].cpu().numpy()
            alt["max_rel_dev_vs_oracle"] = float(max((np.abs(got2 - want) / np.maximum(np.abs(want), scale)).max(),
                                                     (np.abs(gotx2 - wantx) / np.maximum(np.abs(wantx), scale)).max()))
        state["flags"] = mode_flags[args.arith]

    # ---- the other scaling mode (more than one rank only) ------------------------------------------------
    other_scaling = None
    if do_other:
        state["rows"] = rows_other
        o_steps = max(3, args.steps // 2)
        o_el, o_f, o_i, o_total = timed(o_steps, max(1, args.warmup // 2))
        o_val = o_total * n * o_steps / o_el / 1e9
        other_scaling = {"scaling": other, "value": round(o_val, 4), "unit": "Gsamples/s", "steps": o_steps,
                         "signals_total": o_total, "signals_rank0": rows_other, "forward_ms": round(o_f, 4),
                         "inverse_ms": round(o_i, 4),
                         "pct_of_hbm_roofline": round(100.0 * o_val / world * 16 * (J + 2) / HBM_PEAK_GBS, 2)}
        state["rows"] = rows_main

    # the CPU legs run beside the single-GPU line only: with more ranks the other processes would sit waiting for rank 0
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, cpu_soa = cpu_baseline(o, w, J, n, args)

    samples = rows_total * n * args.steps
    value = samples / elapsed / 1e9
    bytes_per_dir = 8.0 * (J + 2) * rows * n
    kern = {}
    for part in plan_txt.split(";"):
        side, _, names = part.strip().partition(": ")
        kern[side] = names
    fwd_name = kern.get("forward", "?").split("[")[0]
    inv_name = kern.get("inverse", "?").split("[")[0]
    dom_name, dom_ms = (inv_name, inv_ms) if inv_ms >= fwd_ms else (fwd_name, fwd_ms)
    achieved = bytes_per_dir / (dom_ms * 1e-3) / 1e9
    # HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/prof_round.sh).  The figure is
    # a measurement of ONE build: it is reported only while the library loaded here is that build (SHA-256 digest),
    # otherwise the line says traffic null / traffic_stale true.
    traffic, traffic_stale = None, False
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            import hashlib
            tj = json.load(open(tpath))
            key = "%s:%dx%d:J%d:%s" % (dom_name, rows, n, J, args.wavelet)
            det = tj.get(key + ":detail") or {}
            have = hashlib.sha256(open(N.LIB_PATH, "rb").read()).hexdigest()[:16]
            if key in tj and det.get("lib_digest") == have:
                traffic = tj[key]
            elif key in tj:
                traffic_stale = True
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
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": workload_text(args, J, rows_total, world, args.scaling),
            "signals_total": rows_total, "signals_rank0": rows, "samples_per_signal": n, "levels": J,
            "wavelet": args.wavelet,
            "parallelism": "rows sharded over %d GPU(s), one process per GPU, no collective" % world,
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
        "other_scaling": other_scaling,
        "roofline": {
            "bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_stale": traffic_stale,
            "algorithmic_bytes_per_launch": bytes_per_dir,
            "plan": plan_txt,
            "other_kernel": {"kernel": fwd_name if inv_ms >= fwd_ms else inv_name,
                             "achieved": round(bytes_per_dir / (min(fwd_ms, inv_ms) * 1e-3) / 1e9, 1)},
        },
        "cpu_baseline": cpu,
        "cpu_baseline_soa": cpu_soa,
    }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(o, w, J, n, args):
    """Times two CPU restatements of the reference on bounded samples of the workload: the per-signal scalar
    path (oracle batch drivers, OpenMP over signals) and the SIMD batch scheme in structure-of-arrays layout."""
    import numpy as np
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
        rows = int(max(probe, min(32768, probe * 10.0 / max(tp, 1e-6))))  # aim at ~10 s
    el = run(rows)
    scalar = {
        "value": round(rows * n / el / 1e9, 5), "unit": "Gsamples/s", "cores": cores, "kind": "port",
        "sample": "%d signals x %d samples, %s J=%d fwd+inv, C restatement of the reference's scalar/vector-order "
                  "Java loops (dense zero-stuffed inverse), gcc -O3 -mavx2 -fopenmp, %.1f s"
                  % (rows, n, args.wavelet.upper(), J, el),
    }
    soa = None
    if args.boundary == 0:
        batch = 256  # signals per structure-of-arrays block: (J+3) planes of batch*n doubles = 302 MB at n=16384

        def run_soa(reps):
            xs = np.ascontiguousarray((base[None, :] + 0.1 * rng.standard_normal((batch, n))).T).reshape(-1)
            t0 = time.perf_counter()
            for _ in range(reps):
                o.soa_multilevel_roundtrip(xs, w, batch, n, J)
            return time.perf_counter() - t0

        tp = run_soa(1)
        reps = int(max(1, min(4096, 8.0 / max(tp, 1e-6))))  # aim at ~8 s
        el2 = run_soa(reps)
        soa = {
            "value": round(reps * batch * n / el2 / 1e9, 5), "unit": "Gsamples/s", "cores": cores, "kind": "port",
            "sample": "%d blocks of %d signals x %d samples, %s J=%d fwd+inv, C restatement of the reference's SIMD "
                      "batch scheme (BatchSIMDMODWT general path per level, structure-of-arrays, lanes along the "
                      "batch, compiler-vectorised AVX2, OpenMP over t), %.1f s" % (reps, batch, n, args.wavelet.upper(), J, el2),
        }
    return scalar, soa


if __name__ == "__main__":
    main()
