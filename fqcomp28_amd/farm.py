"""Block farm: how independent FASTQ blocks are dealt to the GPUs of a node.

The reference's only parallelism is N worker threads pulling whole blocks
(src/process.cpp:46-68, 93-104).  Blocks are independent given the per-archive
frequency tables, so the multi-GPU path is the same thing with one process per
GPU: block b goes to rank b mod world, every rank holds a replica
of the tables, and there is no data-path collective.  torch.distributed is used
for the start/stop barrier and the max-over-ranks of the elapsed time only.
"""


def shard_blocks(n_blocks, rank, world):
    """Indices of the blocks rank `rank` of `world` codes (round-robin, like the
    reference's chunk dispenser when every worker is equally fast)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_blocks, world))


def blocks_for_weak_scaling(blocks_per_gpu, world):
    """Weak scaling: per-GPU work is fixed, so the job has blocks_per_gpu * world blocks."""
    return blocks_per_gpu * world


def reduce_max(value, dist=None):
    """max over ranks of a python float (gloo or nccl backend); identity without dist."""
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return value
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(value, dist=None):
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return value
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
