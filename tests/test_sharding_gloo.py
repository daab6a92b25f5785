"""world_size-2 (and 3) gloo test of the multi-GPU host path on CPU: row partitioning, per-rank
compute on the owned block only, gather to rank 0 in batch order, barrier + MAX-reduce of the
timing as bench.py does.  The per-rank transform is played by the CPU oracle here (no GPU in this
test); on a GPU box the same run_sharded() wraps the library call."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vectorwave_amd.sharding import run_sharded, shard_rows


def test_shard_rows_partition_properties():
    for total in (0, 1, 7, 8, 65536, 65537):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard_rows(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            for a, b in zip(blocks, blocks[1:]):
                assert a[1] == b[0]
            sizes = [b[1] - b[0] for b in blocks]
            assert max(sizes) - min(sizes) <= 1
    # BASELINE config 4: 65536 rows over 8 GPUs -> 8192 rows each
    assert shard_rows(65536, 8, 3) == (3 * 8192, 4 * 8192)
    with pytest.raises(ValueError):
        shard_rows(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, rows, n, out_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as o
    from util import signals
    from vectorwave_amd.wavelets import get_wavelet
    w = get_wavelet("db4")
    x = signals(rows, n, seed=11)  # same seeded batch on every rank
    touched = []

    def compute(block):
        touched.append(block.shape[0])
        st, c = o.decompose_batch(block, w, 3, 0)
        assert st == 0
        return c

    dist.barrier()
    res = run_sharded(compute, x, rank, world)
    # bench.py's timing reduction: MAX over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    r0, r1 = shard_rows(rows, world, rank)
    assert touched == [r1 - r0]
    if rank == 0:
        np.save(out_path, res)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,rows", [(2, 8), (2, 7), (3, 10)])
def test_sharded_forward_matches_single_process(tmp_path, world, rows):
    from oracle import oracle as o
    from util import signals
    from vectorwave_amd.wavelets import get_wavelet
    n = 256
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), rows, n, out), nprocs=world, join=True)
    got = np.load(out)
    st, want = o.decompose_batch(signals(rows, n, seed=11), get_wavelet("db4"), 3, 0)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
