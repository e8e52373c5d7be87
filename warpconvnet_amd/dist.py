"""Data parallelism for the sparse-conv hot path: scenes shard one-per-GPU, gradients are summed by one
flat-bucket all-reduce (RCCL over xGMI on the GPU, gloo in the CPU tests).

The reference has no collective code of its own; scenes never interact (the batch index is
part of the hash key), so the only exchange step of a training iteration is the sum of the weight / bias
gradients.  Those are small (0.88 MB per 64->128 layer), i.e. latency / per-link bound on point-to-point
xGMI: everything is flattened into as few buckets as possible so a step issues one collective, not one per
parameter.  Kernel selection is a pure function of shapes (no run-time autotune), so ranks cannot diverge.
"""
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def rank_and_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_scenes(num_scenes: int, rank: Optional[int] = None, world_size: Optional[int] = None) -> List[int]:
    """Scene i goes to rank i mod W.  Returns the scene ids owned by ``rank``."""
    if rank is None or world_size is None:
        rank, world_size = rank_and_world()
    return list(range(rank, num_scenes, world_size))


@torch.no_grad()
def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, average: bool = True,
                        bucket_bytes: int = 256 << 20) -> int:
    """Sum (or average) ``p.grad`` over the ranks with flat buckets; returns the number of collectives issued.

    Parameters without a gradient contribute zeros so every rank issues identical collectives.
    """
    params = [p for p in params if p.requires_grad]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1 or not params:
        return 0
    world = dist.get_world_size(group)
    calls = 0
    by_dtype = {}
    for p in params:
        by_dtype.setdefault((p.dtype if p.grad is None else p.grad.dtype, p.device), []).append(p)
    for (dtype, device), plist in by_dtype.items():
        bucket: List[torch.nn.Parameter] = []
        size = 0

        def flush(bucket):
            flat = torch.cat([(q.grad if q.grad is not None else torch.zeros_like(q, dtype=dtype)).reshape(-1) for q in bucket])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                flat.div_(world)
            off = 0
            for q in bucket:
                n = q.numel()
                g = flat[off : off + n].view_as(q)
                if q.grad is None:
                    q.grad = g.clone()
                else:
                    q.grad.copy_(g)
                off += n

        for p in plist:
            nbytes = p.numel() * p.element_size()
            if bucket and size + nbytes > bucket_bytes:
                flush(bucket)
                calls += 1
                bucket, size = [], 0
            bucket.append(p)
            size += nbytes
        if bucket:
            flush(bucket)
            calls += 1
    return calls
