"""One-stream-per-GPU sharding helpers: camera streams are independent, so N GPUs run N
shards with NO collective on the data path.  torch.distributed (gloo, CPU tensors) is used only as the
launcher-side plumbing bench.py needs: a barrier, max-over-ranks time, sum of per-rank counts."""
from __future__ import annotations

import os


def env():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(rank: int, world: int):
    """Returns the torch.distributed module (gloo group initialised) or None for a single process."""
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def stream_seeds(rank: int, n: int):
    """Seeds of the synthetic frames of this rank's stream: disjoint across ranks."""
    return [1000 * rank + i for i in range(n)]


def barrier(dist):
    if dist is not None:
        dist.barrier()


def reduce_max(dist, value: float) -> float:
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def reduce_sum(dist, values):
    if dist is None:
        return [float(v) for v in values]
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t]


def gather_floats(dist, value: float, world: int):
    """one float per rank, in rank order (bench.py: per-rank frames/s)"""
    if dist is None:
        return [float(value)]
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(o[0]) for o in out]


def gather_ints(dist, values, world: int):
    """all_gather of a small int list (tests: stream disjointness)."""
    if dist is None:
        return [list(values)]
    import torch
    t = torch.tensor(list(values), dtype=torch.int64)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [[int(v) for v in o] for o in out]


def finish(dist):
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
