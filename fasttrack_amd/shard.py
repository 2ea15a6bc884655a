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


def parse_cpulist(text: str):
    """'0-63,128-191' -> sorted list of CPU numbers (the format of /sys/devices/system/node/node*/cpulist)"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return sorted(set(cpus))


def numa_node_of_pci(bus_id: str, sysfs: str = "/sys") -> int:
    """NUMA node of a PCI device, -1 when the platform does not say"""
    try:
        return int(open(os.path.join(sysfs, "bus/pci/devices", bus_id.lower(), "numa_node")).read().strip())
    except (OSError, ValueError):
        return -1


def pin_to_numa_node(node: int, world: int = 1, slot: int = 0, sysfs: str = "/sys"):
    """Restricts this process (and the threads it starts later: the library's host pool) to the CPUs of `node` that its
    current affinity allows; with several ranks on one node (`world` of them, this one the `slot`-th) every rank takes an
    equal share of those CPUs.  Returns the CPUs chosen, or None when nothing was changed (node unknown, no CPU of the node
    allowed, platform without sched_setaffinity).  On an 8 x MI355X box the GPUs hang off two sockets: a rank whose threads
    marshal frames on the other socket pays the inter-socket hop for every upload."""
    if node < 0 or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        cpus = parse_cpulist(open(os.path.join(sysfs, f"devices/system/node/node{node}/cpulist")).read())
    except OSError:
        return None
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
    if not allowed:
        return None
    if world > 1:
        share = max(1, len(allowed) // world)
        mine = allowed[slot * share:(slot + 1) * share] if slot < world - 1 else allowed[slot * share:]
        allowed = mine or allowed
    os.sched_setaffinity(0, allowed)
    return allowed


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
