"""N>1 path on CPU: two processes over gloo shard one query batch, search their shards independently
(the oracle stands in for the GPU here), and the concatenation equals the single-process result; the
timing reduction is a MAX over ranks.  No data-path collective exists in this design."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from avxwindowfmindex_amd import dist as shard
    from avxwindowfmindex_amd import synth
    from oracle import oracle as O
    r, w = shard.init("gloo")
    assert (r, w) == (rank, world)
    total, K = 10001, 15
    txt = synth.text(21, 60000)
    ix = O.Index.from_text(txt.tobytes(), O.DNA, 8, 6)  # "index replica" of this rank
    begin, end = shard.shard_bounds(total, w, r)
    q = synth.planted_queries(22, end - begin, K, txt, first=begin)  # the shard generates its own slice
    chars, offsets = synth.fixed_csr(q)
    shard.barrier(w)
    sp, ep, cnt, _ = ix.batch_search(chars, offsets)
    slowest = shard.max_over_ranks(1.0 + rank, w)
    assert slowest == float(world)
    allc = shard.gather_counts(cnt, w)
    if r == 0:
        np.save(os.path.join(out_dir, "counts.npy"), allc)
    np.save(os.path.join(out_dir, f"ranges_{rank}.npy"), np.stack([sp, ep], 1))
    dist.destroy_process_group()


def test_two_rank_sharding_over_gloo(tmp_path):
    import torch.multiprocessing as mp
    from avxwindowfmindex_amd import dist as shard
    from avxwindowfmindex_amd import synth
    from oracle import oracle as O
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    total, K = 10001, 15
    txt = synth.text(21, 60000)
    ix = O.Index.from_text(txt.tobytes(), O.DNA, 8, 6)
    chars, offsets = synth.fixed_csr(synth.planted_queries(22, total, K, txt))
    sp, ep, cnt, _ = ix.batch_search(chars, offsets)
    assert np.array_equal(np.load(tmp_path / "counts.npy"), cnt)
    ranges = np.concatenate([np.load(tmp_path / f"ranges_{r}.npy") for r in range(world)])
    assert np.array_equal(ranges[:, 0], sp) and np.array_equal(ranges[:, 1], ep)
    # shards are contiguous, disjoint and cover the batch for any world size
    for w in (1, 2, 3, 8):
        b = [shard.shard_bounds(total, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == total and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
