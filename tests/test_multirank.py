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


def _fallback_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from avxwindowfmindex_amd import dist as shard
    r, w = shard.init("nccl")  # no GPU here: the RCCL trial fails on every rank and all of them agree on gloo
    assert (r, w) == (rank, world) and shard.timing_backend() == "gloo"
    shard.barrier(w)
    assert shard.max_over_ranks(10.0 + rank, w, device="cpu") == 10.0 + world - 1
    parts = shard.gather_objects((rank, "x" * rank), w)
    assert parts == [(i, "x" * i) for i in range(world)]
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("1")
    dist.destroy_process_group()


def test_rccl_failure_falls_back_to_gloo_in_the_same_process(tmp_path):
    """bench.py asks for the RCCL backend; when that cannot be set up (here: no GPU at all) the ranks keep the gloo group
    they rendezvoused on -- no re-exec, no hang -- and the barrier / MAX reduction still work"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the RCCL trial may succeed")
    import torch.multiprocessing as mp
    mp.spawn(_fallback_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_digests_add_up_over_any_sharding_and_catch_errors():
    """the digests bench.py compares across --gpus N: per-k-mer / per-hit hashes keyed by GLOBAL k-mer number, summed"""
    import torch
    from avxwindowfmindex_amd import digest
    rng = np.random.default_rng(3)
    Q = 5000
    counts = rng.integers(0, 4, Q).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    pos = rng.integers(0, 1 << 33, int(off[-1])).astype(np.int64)
    t = torch.from_numpy
    whole = (digest.counts_digest(0, t(counts)), digest.positions_digest(0, t(off), t(pos)))
    for cuts in ([0, Q], [0, 1234, Q], [0, 1, 2, 4000, 4001, Q]):
        dc = dp = 0
        for b, e in zip(cuts, cuts[1:]):
            dc += digest.counts_digest(b, t(counts[b:e]))
            dp += digest.positions_digest(b, t(off[b:e + 1] - off[b]), t(pos[off[b]:off[e]]))
        assert (dc & digest.MASK, dp & digest.MASK) == whole, cuts
    # a count under the wrong k-mer number, two hits of one list swapped, a changed position: all seen
    assert digest.counts_digest(1, t(counts)) != whole[0]
    i = int(np.flatnonzero(counts >= 2)[0])
    swapped = pos.copy()
    swapped[off[i]], swapped[off[i] + 1] = pos[off[i] + 1], pos[off[i]]
    assert pos[off[i]] != pos[off[i] + 1] and digest.positions_digest(0, t(off), t(swapped)) != whole[1]
    # the check against committed values: per shard, as a sum over a chain of committed pieces, unknown, mismatch
    describe = lambda f, c: digest.key("dna", "random", "locate", 1000, "21", 8, 8, f, c)  # noqa: E731
    hexes = lambda dc, dp: {"counts": f"{dc:016x}", "positions": f"{dp:016x}"}  # noqa: E731
    halves = [(0, 2500), (2500, 2500)]
    shard_d = [(b, c, digest.counts_digest(b, t(counts[b:b + c])),
                digest.positions_digest(b, t(off[b:b + c + 1] - off[b]), t(pos[off[b]:off[b + c]]))) for b, c in halves]
    golden_whole = {describe(0, Q): hexes(*whole)}
    assert digest.check_against_golden(shard_d, golden_whole, describe)["status"] == "match"
    golden_halves = {describe(b, c): hexes(dc, dp) for b, c, dc, dp in shard_d}
    assert digest.check_against_golden([(0, Q, *whole)], golden_halves, describe)["status"] == "match"
    assert digest.check_against_golden(shard_d, golden_halves, describe)["status"] == "match"
    assert digest.check_against_golden(shard_d, {}, describe)["status"] == "unknown"
    bad = [shard_d[0], (shard_d[1][0], shard_d[1][1], shard_d[1][2] ^ 1, shard_d[1][3])]
    with pytest.raises(AssertionError):
        digest.check_against_golden(bad, golden_whole, describe)
    with pytest.raises(AssertionError):  # shards that leave a gap
        digest.check_against_golden([shard_d[0], (2600, 2400, 0, 0)], golden_whole, describe)


def test_balanced_shard_bounds_follow_the_weights():
    from avxwindowfmindex_amd import dist as shard
    rng = np.random.default_rng(5)
    lengths = rng.integers(8, 31, 100_000).astype(np.int64)
    prefix = np.concatenate([[0], np.cumsum(lengths)])
    for world in (1, 2, 4, 8):
        b = [shard.balanced_bounds(prefix, world, r) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == lengths.size and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        work = [int(prefix[e] - prefix[s]) for s, e in b]
        assert max(work) - min(work) <= 31, work  # within one k-mer of each other


def _exchange_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from avxwindowfmindex_amd import dist as shard
    r, w = shard.init("gloo")
    buckets = 64
    rng = np.random.default_rng(100 + rank)
    counts = rng.integers(0, 50, buckets)
    counts[rng.integers(0, buckets, 9)] = 0  # empty buckets too
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    # a record that says where it comes from: bucket << 40 | rank << 32 | its place in the bucket
    recs = np.concatenate([(np.int64(b) << 40) | (np.int64(rank) << 32) | np.arange(counts[b], dtype=np.int64) for b in range(buckets)] or [np.zeros(0, np.int64)])
    mine, mstart = shard.bucket_exchange(torch.from_numpy(recs), start, buckets, w, r)
    cuts = shard.bucket_cuts(buckets, w)
    np.save(os.path.join(out_dir, f"sent_{rank}.npy"), recs)
    np.save(os.path.join(out_dir, f"got_{rank}.npy"), mine.numpy())
    np.save(os.path.join(out_dir, f"start_{rank}.npy"), mstart.numpy())
    assert len(mstart) == cuts[r + 1] - cuts[r] + 1 and int(mstart[-1]) == mine.numel()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_seed_bucket_exchange_over_gloo(tmp_path, world):
    """round 6, the one exchange of the seed-bucket sharding (dist.bucket_exchange; include/awfm_gpu.h: awfmGpuOrderKmers): every
    rank ends up with exactly the records of its bucket range from all ranks, bucket by bucket, with the bucket starts that go
    with them -- here on host tensors over gloo; the records say where they came from"""
    import torch.multiprocessing as mp
    from avxwindowfmindex_amd import dist as shard
    mp.spawn(_exchange_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    buckets = 64
    cuts = shard.bucket_cuts(buckets, world)
    sent = np.concatenate([np.load(tmp_path / f"sent_{r}.npy") for r in range(world)])
    for r in range(world):
        got, start = np.load(tmp_path / f"got_{r}.npy"), np.load(tmp_path / f"start_{r}.npy")
        b = got >> 40
        assert np.all(np.diff(b) >= 0) and (len(b) == 0 or (b.min() >= cuts[r] and b.max() < cuts[r + 1])), "not in bucket order / another rank's bucket"
        want = np.sort(sent[((sent >> 40) >= cuts[r]) & ((sent >> 40) < cuts[r + 1])])
        assert np.array_equal(np.sort(got), want), "records lost, duplicated or sent to the wrong rank"
        for i, bucket in enumerate(range(cuts[r], cuts[r + 1])):
            assert np.all(b[start[i]:start[i + 1]] == bucket)
