"""bench.py's own launcher on CPU: `--gpus N` without an outer launcher must start N rank processes, shard the
rows over them, reduce the timing over ranks and print one line from rank 0; a failing rank must fail the job.
The ranks run in --dry-run mode (gloo rendezvous, a stub instead of the transform): no GPU is touched."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + list(argv), env=env, capture_output=True, text=True, timeout=300)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 3])
def test_gpus_flag_starts_that_many_ranks_strong_scaled(world):
    p = _run("--gpus", str(world), "--steps", "2", "--warmup", "0", "--rows", "65536", "--dry-run")
    assert p.returncode == 0, p.stderr
    out = _json_line(p.stdout)
    assert out["n_gpus"] == world
    assert out["scaling"] == "strong"
    # the 65536 rows of the job are split, not replicated
    assert out["config"]["signals_total"] == 65536
    assert out["config"]["signals_rank0"] == 65536 // world
    assert out["ms_per_step"] >= 1.0 * world  # MAX over ranks: the stub of rank r sleeps (r+1) ms per step


def test_weak_scaling_flag_keeps_the_batch_per_rank():
    p = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--rows", "1000", "--scaling", "weak", "--dry-run")
    assert p.returncode == 0, p.stderr
    out = _json_line(p.stdout)
    assert out["scaling"] == "weak" and out["config"]["signals_total"] == 2000 and out["config"]["signals_rank0"] == 1000


def test_failing_rank_fails_the_job_and_prints_no_line():
    p = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run", "--dry-run-fail-rank", "1")
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "rank 1 of 2 exited" in p.stderr


def test_gpus_must_match_the_outer_launcher():
    p = _run("--gpus", "2", "--dry-run", env_extra={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert p.returncode != 0 and "launcher started 1 rank" in p.stderr


def test_more_ranks_than_gpus_fails_loudly():
    """Without --dry-run every rank needs its own GPU: on a box with fewer (here: none) the job must fail, never
    fall back to fewer ranks."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    p = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_soa_batch_baseline_equals_per_signal_oracle():
    """The second CPU baseline (SIMD batch scheme, structure-of-arrays) performs the per-signal path's operations in
    the same order: its coefficients and reconstruction equal the scalar oracle's bit for bit."""
    from oracle import oracle as o
    from util import assert_bits_equal, signals
    from vectorwave_amd.wavelets import get_wavelet
    for wname, b, n, J in (("db4", 5, 300, 4), ("haar", 3, 64, 5), ("sym8", 4, 1000, 3)):
        w = get_wavelet(wname)
        x = signals(b, n, seed=3)
        soa = o.convert_to_soa(x)
        c, xr = o.soa_multilevel_roundtrip(soa, w, b, n, J)
        st, want = o.decompose_batch(x, w, J, 0)
        assert st == 0
        got = np.stack([o.convert_from_soa(c[k], b, n) for k in range(J + 1)], axis=1)
        assert_bits_equal(got, want, "%s soa coefficients" % wname)
        assert_bits_equal(o.convert_from_soa(xr, b, n), o.reconstruct_batch(want, w, 0), "%s soa reconstruction" % wname)


@pytest.mark.gpu
def test_gpus_2_on_a_one_gpu_box_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with exactly one GPU")
    p = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--rows", "64", "--n", "1024", "--no-cpu-baseline")
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "needs GPU 1 but this node shows 1 GPU" in p.stderr
