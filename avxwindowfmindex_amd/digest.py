"""Additive digests of batch results, so that any sharding of a batch can be checked against one committed value.

A batch's results are per k-mer: a count, and -- locate -- a list of positions in BWT order (what
awFmParallelSearchCount/Locate leave in each AwFmKmerSearchData, ref src/AwFmParallelSearch.c:159-220, :315-365).
The digest of a shard is the SUM (mod 2^64) of one 64-bit hash per k-mer / per hit, keyed by the k-mer's GLOBAL number
in the batch and the hit's rank in its list:

    counts    : sum_i  mix(A * (first + i) + B * count_i)
    positions : sum_h  mix(A * (first + query(h)) + C * rank(h) + D * position_h)

A sum does not care how the batch was cut, so the digests of N ranks' shards add up to the digest of the 1-rank run --
which is what bench.py checks at --gpus N against the values committed in tests/golden/bench_digests.json -- while a
k-mer answered under the wrong number, a list in the wrong order or a missing hit all change it.
Computed with torch on whatever device the results live on (int64 arithmetic wraps; the shifts are arithmetic, which is
as good a mixer as any as long as every implementation agrees -- there is only this one).
"""
import json
import os

_A, _B, _C, _D = 0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0xD6E8FEB86659FD93
MASK = (1 << 64) - 1
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bench_digests.json")


def _s64(c):
    return c - (1 << 64) if c >= (1 << 63) else c


def _mix(x):
    x = (x ^ (x >> 30)) * _s64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> 27)) * _s64(0x94D049BB133111EB)
    return x ^ (x >> 31)


def counts_digest(first, counts):
    """counts: 1-D integer torch tensor (the 32-bit counts of k-mers first .. first+len-1)"""
    import torch
    total = 0
    n = counts.numel()
    step = 1 << 24
    for b in range(0, n, step):
        c = counts[b:b + step].to(torch.int64) & 0xFFFFFFFF
        ids = torch.arange(first + b, first + b + c.numel(), dtype=torch.int64, device=counts.device)
        total += int(_mix(ids * _s64(_A) + c * _s64(_B)).sum().item())
    return total & MASK


def positions_digest(first, hit_offsets, positions):
    """hit_offsets: int64[n+1] CSR offsets of the shard's k-mers, positions: int64[hit_offsets[n]] in list order"""
    import torch
    n = hit_offsets.numel() - 1
    total = 0
    step = 1 << 23
    for b in range(0, n, step):
        e = min(n, b + step)
        off = hit_offsets[b:e + 1]
        lo, hi = int(off[0].item()), int(off[-1].item())
        if hi == lo:
            continue
        lens = off[1:] - off[:-1]
        q = torch.repeat_interleave(torch.arange(first + b, first + e, dtype=torch.int64, device=off.device), lens)
        starts = torch.repeat_interleave(off[:-1], lens)
        rank = torch.arange(lo, hi, dtype=torch.int64, device=off.device) - starts
        p = positions[lo:hi].to(torch.int64)
        total += int(_mix(q * _s64(_A) + rank * _s64(_C) + p * _s64(_D)).sum().item())
    return total & MASK


def counts_digest_keyed(ids, counts):
    """the same sum over k-mers that are named one by one (ids: their numbers in the whole batch, any order) -- what a rank of
    a seed-bucket-sharded run holds: a share of the ORDER, not a stretch of the batch"""
    import torch
    total = 0
    step = 1 << 24
    for b in range(0, ids.numel(), step):
        c = counts[b:b + step].to(torch.int64) & 0xFFFFFFFF
        total += int(_mix(ids[b:b + step].to(torch.int64) * _s64(_A) + c * _s64(_B)).sum().item())
    return total & MASK


def positions_digest_keyed(ids, hit_offsets, positions):
    """positions_digest for entries named one by one: entry e is k-mer ids[e], its hits positions[hit_offsets[e]:hit_offsets[e + 1]]"""
    import torch
    n = hit_offsets.numel() - 1
    total = 0
    step = 1 << 23
    for b in range(0, n, step):
        e = min(n, b + step)
        off = hit_offsets[b:e + 1]
        lo, hi = int(off[0].item()), int(off[-1].item())
        if hi == lo:
            continue
        lens = off[1:] - off[:-1]
        q = torch.repeat_interleave(ids[b:e].to(torch.int64), lens)
        starts = torch.repeat_interleave(off[:-1], lens)
        rank = torch.arange(lo, hi, dtype=torch.int64, device=off.device) - starts
        p = positions[lo:hi].to(torch.int64)
        total += int(_mix(q * _s64(_A) + rank * _s64(_C) + p * _s64(_D)).sum().item())
    return total & MASK


def key(alphabet, workload, mode, text_len, kmer, seed_k, sa_ratio, first, count):
    """the name of a committed digest: the synthetic inputs are functions of these and of fixed seeds (bench.py)"""
    return f"{alphabet}:{workload}:{mode}:n{text_len}:k{kmer}:seed{seed_k}:ratio{sa_ratio}:first{first}:count{count}"


def load_golden(path=None):
    path = path or os.environ.get("AWFM_BENCH_DIGESTS") or GOLDEN  # the override: tests that make their own 1-rank digests
    if not os.path.exists(path):
        return {}
    return json.load(open(path))


def check_against_golden(entries, golden, describe):
    """entries: [(first, count, counts_digest, positions_digest or None)] of all ranks; `describe(first, count)` -> key.

    The shards must tile [lo, hi).  Compared, in this order of preference: each shard with its own committed entry;
    the sum of all shards with a committed entry (or a chain of committed entries) covering [lo, hi).
    Returns {"status": "match" | "unknown", ...}; raises AssertionError on a mismatch or on shards that do not tile."""
    entries = sorted(entries)
    for (f0, c0, *_), (f1, *_rest) in zip(entries, entries[1:]):
        assert f0 + c0 == f1, f"shards do not tile the batch: [{f0}, {f0 + c0}) then {f1}"
    lo, hi = entries[0][0], entries[-1][0] + entries[-1][1]
    total_c = sum(e[2] for e in entries) & MASK
    has_p = all(e[3] is not None for e in entries)
    total_p = (sum(e[3] for e in entries) & MASK) if has_p else None
    out = {"range": [lo, hi], "shards": len(entries), "counts": f"{total_c:016x}",
           "positions": f"{total_p:016x}" if has_p else None}

    def committed(first, count):
        g = golden.get(describe(first, count))
        return (int(g["counts"], 16), int(g["positions"], 16) if g.get("positions") else None) if g else None

    per_shard = [committed(f, c) for f, c, *_ in entries]
    if all(per_shard):
        for (f, c, dc, dp), (gc, gp) in zip(entries, per_shard):
            assert dc == gc, f"shard [{f}, {f + c}): counts digest {dc:016x} != committed {gc:016x} (1-rank run)"
            assert dp is None or gp is None or dp == gp, f"shard [{f}, {f + c}): positions digest {dp:016x} != committed {gp:016x}"
        out.update(status="match", compared="every shard with its committed 1-rank digest")
        return out
    # a chain of committed entries that tiles [lo, hi): sums are additive
    chain, at = [], lo
    by_first = {}
    for k, v in golden.items():
        prefix = describe(0, 0).rsplit(":first", 1)[0]
        if k.startswith(prefix + ":first"):
            f, c = k[len(prefix) + 6:].split(":count")
            by_first.setdefault(int(f), []).append((int(c), v))
    while at < hi:
        options = [o for o in by_first.get(at, []) if at + o[0] <= hi]
        if not options:
            break
        c, v = max(options, key=lambda o: o[0])
        chain.append(v)
        at += c
    if at == hi and chain:
        gc = sum(int(v["counts"], 16) for v in chain) & MASK
        assert total_c == gc, f"batch [{lo}, {hi}): counts digest {total_c:016x} != committed {gc:016x} (sum of {len(chain)} 1-rank runs)"
        if has_p and all(v.get("positions") for v in chain):
            gp = sum(int(v["positions"], 16) for v in chain) & MASK
            assert total_p == gp, f"batch [{lo}, {hi}): positions digest {total_p:016x} != committed {gp:016x}"
        out.update(status="match", compared=f"sum over the ranks with the sum of {len(chain)} committed 1-rank digest(s)")
        return out
    out.update(status="unknown", compared="no committed digest covers this batch")
    return out
