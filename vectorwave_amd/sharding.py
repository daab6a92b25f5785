"""Row sharding of a signal batch over the GPUs of one node.

Signals are independent (the reference's batch loop carries no state between
rows: modwt/MODWTTransform.java:564-567), so the multi-GPU layout is plain
partitioning: one process per GPU, rank g of G owns the contiguous row block
[g*B//G, (g+1)*B//G), transforms it with its own library context, and the
results are gathered to the host of rank 0.  There is no data-path collective;
torch.distributed is used only to move the finished host arrays (gloo) and for
barriers.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_rows(total_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of rows owned by `rank`: [r0, r1).  Blocks differ by at most one row
    in size when world_size does not divide total_rows, cover every row once, and stay in
    order so that concatenating the ranks' outputs restores the batch order."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size %d/%d" % (rank, world_size))
    if total_rows < 0:
        raise ValueError("negative row count")
    r0 = (rank * total_rows) // world_size
    r1 = ((rank + 1) * total_rows) // world_size
    return r0, r1


def run_sharded(compute: Callable[[np.ndarray], np.ndarray], x: np.ndarray, rank: int = 0, world_size: int = 1,
                gather: bool = True, group=None) -> Optional[np.ndarray]:
    """Apply `compute` (rows [b][N] -> [b][...]) to this rank's block of `x` and gather the
    blocks on rank 0 in row order.  `x` is the full batch (every rank sees the same host
    array, e.g. a memory-mapped file); only the owned rows are touched.  Returns the full
    result on rank 0 (None elsewhere) when gather is True, else the local block."""
    r0, r1 = shard_rows(x.shape[0], world_size, rank)
    local = compute(np.ascontiguousarray(x[r0:r1]))
    if not gather:
        return local
    if world_size == 1:
        return local
    import torch
    import torch.distributed as dist

    # shapes differ between ranks when world_size does not divide B: exchange row counts first
    mine = torch.from_numpy(np.ascontiguousarray(local))
    tail = tuple(mine.shape[1:])
    if rank == 0:
        parts = []
        for src in range(world_size):
            s0, s1 = shard_rows(x.shape[0], world_size, src)
            parts.append(torch.empty((s1 - s0,) + tail, dtype=mine.dtype))
        parts[0].copy_(mine)
        reqs = [dist.irecv(parts[src], src=src, group=group) for src in range(1, world_size) if parts[src].numel()]
        for r in reqs:
            r.wait()
        return torch.cat(parts, dim=0).numpy()
    if mine.numel():
        dist.send(mine, dst=0, group=group)
    return None
