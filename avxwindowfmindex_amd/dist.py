"""Query sharding across ranks (one process per GPU, index replicated, no data-path collective).

The reference's only parallel axis is OpenMP over independent 8-query blocks
(ref src/AwFmParallelSearch.c:103-129); across GPUs the same independence lets every rank take a
contiguous shard of the batch and write a disjoint slice of the results.  torch.distributed is used
for launch plumbing only: a barrier around the timed region and a MAX over the ranks' wall times.
"""
import os


def env_world():
    """(rank, world_size, local_rank) as torchrun exports them; single process when absent"""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_bounds(total, world, rank):
    """contiguous shard [begin, end) of `total` queries for `rank`: ceil(total/world) per rank"""
    per = (total + world - 1) // world
    begin = min(total, rank * per)
    return begin, min(total, begin + per)


def init(backend):
    """initialise torch.distributed when launched with WORLD_SIZE > 1; returns (rank, world)"""
    import torch.distributed as dist
    rank, world, local_rank = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if backend == "nccl":
            import torch
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world


def barrier(world, sync=None):
    import torch.distributed as dist
    if sync is not None:
        sync()
    if world > 1:
        dist.barrier()
    if sync is not None:
        sync()


def max_over_ranks(value, world, device="cpu"):
    """MAX of a python float over all ranks (the job's step time is the slowest rank's)"""
    if world == 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(local, world):
    """all ranks' 1-D numpy count arrays concatenated in rank order (verification helper, not on the timed path)"""
    if world == 1:
        return local
    import numpy as np
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, local)
    return np.concatenate(parts)
