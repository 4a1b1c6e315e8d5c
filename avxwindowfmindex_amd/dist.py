"""Query sharding across ranks (one process per GPU, index replicated; no data-path collective -- but for the seed-bucket
sharding of dense-hit batches, whose one exchange of records is below: bucket_exchange).

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


_TIMING_GROUP = None   # the group the barrier and the MAX reduction run on (None: the default group)
_TIMING_BACKEND = None  # "nccl" | "gloo" once initialised with world > 1


def timing_backend():
    return _TIMING_BACKEND


def timing_group():
    """the process group device tensors travel on (the RCCL group when it could be set up; None: the default gloo group)"""
    return _TIMING_GROUP


def init(backend):
    """initialise torch.distributed when launched with WORLD_SIZE > 1; returns (rank, world).

    The default group is always gloo (TCP on 127.0.0.1: it cannot fail for GPU reasons).  With backend "nccl" an RCCL
    group over the same ranks is tried next -- created, exercised with one all-reduce under a timeout, and adopted for
    the barrier / MAX reduction only if EVERY rank says it worked (agreement over gloo); otherwise all ranks stay on
    gloo, in this same process (nothing is re-executed).  The data path has no collective either way."""
    global _TIMING_GROUP, _TIMING_BACKEND
    import torch.distributed as dist
    rank, world, local_rank = env_world()
    if world > 1 and not dist.is_initialized():
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=1800))
        _TIMING_BACKEND = "gloo"
        if backend == "nccl":
            import torch
            ok, why = 1, ""
            try:
                # errors and time-outs of the trial all-reduce must surface as exceptions here, not abort the process
                os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
                torch.cuda.set_device(local_rank)
                group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
                probe = torch.ones(1, device=torch.device("cuda", local_rank))
                dist.all_reduce(probe, group=group)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    ok, why = 0, f"trial all-reduce returned {probe.item()}"
            except Exception as e:  # noqa: BLE001 -- anything RCCL / HIP raises: fall back
                ok, why = 0, f"{type(e).__name__}: {e}"
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # over gloo
            if int(flag.item()) == 1:
                _TIMING_GROUP, _TIMING_BACKEND = group, "nccl"
            elif rank == 0 or not ok:
                import sys
                sys.stderr.write(f"[dist] rank {rank}: RCCL group not usable ({why or 'another rank failed'}); "
                                 "barrier and MAX reduction stay on gloo\n")
    return rank, world


def barrier(world, sync=None):
    import torch.distributed as dist
    if sync is not None:
        sync()
    if world > 1:
        dist.barrier(group=_TIMING_GROUP)
    if sync is not None:
        sync()


def max_over_ranks(value, world, device="cpu"):
    """MAX of a python float over all ranks (the job's step time is the slowest rank's)"""
    if world == 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device if _TIMING_BACKEND == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_TIMING_GROUP)
    return float(t.item())


def gather_objects(obj, world):
    """every rank's python object, in rank order (over the gloo default group; verification, not on the timed path)"""
    if world == 1:
        return [obj]
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, obj)
    return parts


def balanced_bounds(prefix_weights, world, rank):
    """contiguous shard [begin, end) of items whose weights have the inclusive-exclusive prefix sums `prefix_weights`
    (len n+1, prefix_weights[0] == 0): cut where the running weight passes rank/world of the total, so that shards of a
    batch of unequal items (mixed-length k-mers: work grows with length) finish together"""
    n = len(prefix_weights) - 1
    total = int(prefix_weights[-1])
    if type(prefix_weights).__module__.startswith("torch"):  # a torch tensor (bench.py keeps the 10^8 lengths on the GPU)
        import torch

        def find(v):
            return int(torch.searchsorted(prefix_weights, torch.tensor([v], dtype=prefix_weights.dtype,
                                                                       device=prefix_weights.device)).item())
    else:
        import numpy as np

        def find(v):
            return int(np.searchsorted(prefix_weights, v, side="left"))

    def cut(r):
        return 0 if r <= 0 else (n if r >= world else min(find((total * r) // world), n))
    return cut(rank), cut(rank + 1)


def gather_counts(local, world):
    """all ranks' 1-D numpy count arrays concatenated in rank order (verification helper, not on the timed path)"""
    if world == 1:
        return local
    import numpy as np
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, local)
    return np.concatenate(parts)


# ---- seed-bucket sharding (round 6; include/awfm_gpu.h: awfmGpuOrderKmers / awfmGpuSearchOrderedRecords) ----

def bucket_cuts(buckets, world):
    """bucket ranges of the ranks: rank r takes [cuts[r], cuts[r + 1])"""
    return [(buckets * r) // world for r in range(world + 1)]


def merge_bucket_slices(slices, starts, first_bucket, end_bucket):
    """What a rank holds after the exchange, put in bucket order.  slices[j]: the records rank j sent (its buckets
    [first_bucket, end_bucket), bucket by bucket); starts[j]: rank j's bucket starts over those buckets, relative to the slice
    (end_bucket - first_bucket + 1 numbers).  Returns (records, bucket starts of the merged array, absolute bucket numbers): the
    runs of one bucket from all ranks end up next to each other, rank by rank -- any order inside a bucket will do
    (awfm_ordered_kernel.h).  torch tensors (any device) or numpy arrays."""
    import torch
    nb = end_bucket - first_bucket
    starts = [torch.as_tensor(s, dtype=torch.int64) for s in starts]
    counts = torch.stack([s[1:] - s[:-1] for s in starts])  # [rank][bucket]
    per_bucket = counts.sum(0)
    merged_start = torch.zeros(nb + 1, dtype=torch.int64)
    merged_start[1:] = torch.cumsum(per_bucket, 0)
    before = torch.cumsum(counts, 0) - counts  # records of the bucket from the ranks before j
    device = slices[0].device if hasattr(slices[0], "device") else "cpu"
    out = torch.empty(int(merged_start[-1]), dtype=torch.int64, device=device)
    for j, sl in enumerate(slices):
        sl = torch.as_tensor(sl)
        if sl.numel() == 0:
            continue
        # destination of record p of slice j: merged_start[b] + before[j][b] + (p - starts[j][b]), b its bucket
        shift = (merged_start[:-1] + before[j] - starts[j][:-1]).to(device)
        reps = counts[j].to(device)
        dest = torch.arange(sl.numel(), dtype=torch.int64, device=device) + torch.repeat_interleave(shift, reps)
        out[dest] = sl.view(torch.int64) if sl.dtype != torch.int64 else sl
    return out, merged_start


def bucket_exchange(records, bucket_start, buckets, world, rank, group=None):
    """The one exchange of the seed-bucket sharding: every rank sends rank j the records of its buckets
    [cuts[j], cuts[j + 1]) -- a contiguous slice of its bucket-ordered array -- and puts what it receives in bucket order
    (merge_bucket_slices).  `records`: int64 tensor (device or host), `bucket_start`: the buckets + 1 first words
    awfmGpuOrderKmers left (records before every bucket).  Returns (records of this rank's buckets in bucket order, their
    bucket starts relative to that array).  The transport is torch.distributed's all_to_all_single on the tensors' device
    (RCCL over xGMI for device tensors under the nccl backend; gloo moves host tensors -- the tests' way)."""
    import torch
    import torch.distributed as dist
    cuts = bucket_cuts(buckets, world)
    bs = torch.as_tensor(bucket_start, dtype=torch.int64).cpu()
    send_at = [int(bs[c]) for c in cuts]
    send_sizes = [send_at[j + 1] - send_at[j] for j in range(world)]
    rel = [(bs[cuts[j]: cuts[j + 1] + 1] - bs[cuts[j]]).tolist() for j in range(world)]  # my starts inside the slice for rank j
    if world == 1:
        return records[send_at[0]: send_at[1]], torch.tensor(rel[0], dtype=torch.int64)
    # sizes and per-bucket starts first (small, over the default gloo group), then the records
    theirs = [None] * world
    dist.all_gather_object(theirs, rel)  # theirs[j][r]: rank j's starts inside the slice it sends to rank r
    mine = [torch.tensor(theirs[j][rank], dtype=torch.int64) for j in range(world)]
    recv_sizes = [int(m[-1]) for m in mine]
    send = records[send_at[0]: send_at[-1]].contiguous()
    recv = torch.empty(sum(recv_sizes), dtype=records.dtype, device=records.device)
    dist.all_to_all_single(recv, send, output_split_sizes=recv_sizes, input_split_sizes=send_sizes, group=group)
    parts, at = [], 0
    for n in recv_sizes:
        parts.append(recv[at: at + n])
        at += n
    return merge_bucket_slices(parts, mine, cuts[rank], cuts[rank + 1])


def bucket_exchange_on_device(g, records, bucket_start, buckets, world, rank, group=None, stream=0):
    """bucket_exchange for records that live on the device of the image `g` (api.GpuIndex): the same all-to-all, and the
    slices put in bucket order by ONE kernel (awfmGpuMergeBucketRuns) instead of torch's index arithmetic.  Returns (records of
    this rank's buckets in bucket order, the buckets + 3 bucket starts awfmGpuSearchOrderedRecords wants, on the device)."""
    import torch
    import torch.distributed as dist
    cuts = bucket_cuts(buckets, world)
    bs = torch.as_tensor(bucket_start, dtype=torch.int64).cpu()
    send_at = [int(bs[c]) for c in cuts]
    send_sizes = [send_at[j + 1] - send_at[j] for j in range(world)]
    rel = [(bs[cuts[j]: cuts[j + 1] + 1] - bs[cuts[j]]).tolist() for j in range(world)]
    if world == 1:
        theirs_for_me, recv, recv_sizes = [rel[0]], records[send_at[0]: send_at[1]], [send_sizes[0]]
    else:
        theirs = [None] * world
        dist.all_gather_object(theirs, rel)  # theirs[j][r]: rank j's starts inside the slice it sends to rank r
        theirs_for_me = [theirs[j][rank] for j in range(world)]
        recv_sizes = [int(t[-1]) for t in theirs_for_me]
        send = records[send_at[0]: send_at[-1]].contiguous()
        recv = torch.empty(sum(recv_sizes), dtype=records.dtype, device=records.device)
        dist.all_to_all_single(recv, send, output_split_sizes=recv_sizes, input_split_sizes=send_sizes, group=group)
    slice_at = torch.tensor([sum(recv_sizes[:j]) for j in range(len(recv_sizes))], dtype=torch.int64)
    starts = torch.tensor(theirs_for_me, dtype=torch.int32)  # [slice][bucket of mine + 1]
    d_slice_at, d_starts = slice_at.to(records.device, non_blocking=True), starts.to(records.device, non_blocking=True)
    out = torch.empty(max(int(sum(recv_sizes)), 1), dtype=torch.int64, device=records.device)
    d_full = torch.empty(buckets + 3, dtype=torch.int32, device=records.device)
    g.merge_bucket_runs(recv.data_ptr(), d_slice_at.data_ptr(), d_starts.data_ptr(), len(recv_sizes), cuts[rank], cuts[rank + 1], buckets,
                        out.data_ptr(), d_full.data_ptr(), stream)
    keep = (recv, d_slice_at, d_starts)  # (alive until the kernel has run: the caller keeps the tuple)
    return out[: int(sum(recv_sizes))], d_full, keep


def full_bucket_start(merged_start, first_bucket, end_bucket, buckets):
    """the bucket starts awfmGpuSearchOrderedRecords wants (buckets + 3 words) for an array that holds the buckets
    [first_bucket, end_bucket) only: nothing before them, everything before what follows them"""
    import torch
    total = int(merged_start[-1])
    out = torch.zeros(buckets + 3, dtype=torch.int64)
    out[first_bucket: end_bucket + 1] = torch.as_tensor(merged_start, dtype=torch.int64)
    out[end_bucket + 1: buckets + 2] = total
    return out.to(torch.int32)
