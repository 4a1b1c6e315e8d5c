#!/usr/bin/env python3
"""Differential fuzz of the ordered hits-only search (pair steps) and of the exact general search with pair steps
against the general kernel stepping letter by letter, and of the pair-step LF walk against the one-letter walk, all on
the GPU: random index sizes, seed depths, deeper tables,
fixed and mixed k-mer lengths, ambiguity characters and runs (flagged pair blocks), buffer alignments.
With FUZZ_WIDE=1 the side under test (hits-only search, second locate) runs the 64-bit-position instantiations
while the general kernel it is compared with keeps 32-bit positions.
usage: scripts/fuzz_ordered.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from avxwindowfmindex_amd import _lib, api  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
L = _lib.lib()
dev = torch.device("cuda")
fuzz_wide = os.environ.get("FUZZ_WIDE") == "1"
t_end = time.time() + budget
rounds = 0
while time.time() < t_end:
    n = int(rng.integers(200_000, 40_000_000))
    seed_k = int(rng.integers(2, 13))
    while 4 ** seed_k > 8 * n:
        seed_k -= 1
    deep_k = int(seed_k + rng.integers(1, 4)) if rng.random() < 0.4 else 0
    ratio = int(rng.choice([1, 3, 8, 16, 255]))
    d_text = torch.empty(n, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthText(d_text.data_ptr(), 0, n, int(rng.integers(1, 1 << 30)), 0, None) == 1
    if rng.random() < 0.5:  # ambiguity runs in the text: their blocks, and those their LF images fall into, are flagged
        for _ in range(int(rng.integers(1, 30))):
            at = int(rng.integers(0, n - 100))
            d_text[at:at + int(rng.integers(1, 90))] = ord("n")
    ix = api.gpu_create_index(d_text.data_ptr(), api.AwFmAlphabetDna, ratio, seed_k, on_device_length=n)
    g = api.GpuIndex(ix, acquire=True)
    g.set_ordered(1)
    if deep_k and 4 ** deep_k * 16 < 2 << 30:
        g.set_deep_seed(deep_k)
    else:
        deep_k = 0
    for _ in range(3):
        Q = int(rng.integers(50_000, 3_000_000))
        qseed = int(rng.integers(1, 1 << 30))
        mis = int(rng.integers(0, 4))
        if rng.random() < 0.5:
            lo, hi = sorted(int(x) for x in rng.integers(1, 41, size=2))
            d_len = torch.empty(Q, dtype=torch.int64, device=dev)
            assert L.awfmGpuSynthMixedLengths(d_len.data_ptr(), 0, Q, lo, hi, qseed, None) == 1
            d_off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
            torch.cumsum(d_len, 0, out=d_off[1:])
            total = int(d_off[-1])
            buf = torch.zeros(total + 16, dtype=torch.uint8, device=dev)
            d_off += mis  # offsets relative to an aligned base: shifts every k-mer by `mis` bytes
            assert L.awfmGpuSynthMixedQueries(buf.data_ptr(), d_off.data_ptr(), 0, Q, qseed, d_text.data_ptr(), n, 0, None) == 1
            chars_ptr, off_ptr, K, desc = buf.data_ptr(), d_off.data_ptr(), 0, f"csr {lo}..{hi}"
            nchars = total + mis
        else:
            K = int(rng.integers(1, 36))
            buf = torch.zeros(Q * K + 16, dtype=torch.uint8, device=dev)
            half = Q // 2
            assert L.awfmGpuSynthRandomQueries(buf.data_ptr() + mis, 0, half, K, qseed, 0, None) == 1
            if K <= n:
                assert L.awfmGpuSynthPlantedQueries(buf.data_ptr() + mis + half * K, half, Q - half, K, qseed + 1,
                                                    d_text.data_ptr(), n, None) == 1
            chars_ptr, off_ptr, desc = buf.data_ptr() + mis, 0, f"fixed {K}"
            nchars = Q * K + mis
        if rng.random() < 0.5:  # ambiguity characters and upper case in the queries
            where = torch.rand(nchars, device=dev)
            buf[:nchars][where < 0.001] = ord("x")
            up = (where > 0.7) & (buf[:nchars] >= ord("a"))
            buf[:nchars][up] -= 32
        exact = torch.zeros(Q * 2, dtype=torch.int64, device=dev)
        hits = torch.full((Q * 2,), 9, dtype=torch.int64, device=dev)
        counts = torch.full((Q,), 9, dtype=torch.int32, device=dev)
        g.set_kernel(api.AWFM_GPU_KERNEL_GROUP2)  # the reference side: the general kernel, two lanes per k-mer, one letter per step, no table beyond the index's
        g.search(chars_ptr, off_ptr, K, Q, exact.data_ptr(), 0)
        torch.cuda.synchronize()
        g.set_kernel(api.AWFM_GPU_KERNEL_AUTO)
        g.set_wide(fuzz_wide)
        # the exact search with pair steps: every range, the empty ones of k-mers without hits included
        exact_pair = torch.full((Q * 2,), 5, dtype=torch.int64, device=dev)
        g.search(chars_ptr, off_ptr, K, Q, exact_pair.data_ptr(), 0)
        torch.cuda.synchronize()
        if not torch.equal(exact, exact_pair):
            print(f"MISMATCH exact pair search n={n} k={seed_k} deep={deep_k} ratio={ratio} Q={Q} {desc}", flush=True)
            sys.exit(1)
        assert g.is_wide == fuzz_wide
        # lookup first (the kernel that looks the deeper table up also searches what is still alive): forced, never, or by
        # its sample -- whichever applies to this batch and image
        lookup = str(rng.choice(["1", "0", ""]))
        if lookup:
            os.environ["AWFM_GPU_LOOKUP_FIRST"] = lookup
        else:
            os.environ.pop("AWFM_GPU_LOOKUP_FIRST", None)
        # mixed-length batches: the lookup-first kernel over the tables per k-mer length, forced / never / by its sample
        mixed = str(rng.choice(["1", "0", ""]))
        if mixed:
            os.environ["AWFM_GPU_MIXED_LOOKUP"] = mixed
        else:
            os.environ.pop("AWFM_GPU_MIXED_LOOKUP", None)
        # lookup prediction (one front end launched when earlier verdicts agree) on or off
        if rng.random() < 0.3:
            os.environ["AWFM_GPU_LOOKUP_PREDICT"] = "0"
        else:
            os.environ.pop("AWFM_GPU_LOOKUP_PREDICT", None)
        g.search_hits(chars_ptr, off_ptr, K, Q, hits.data_ptr(), counts.data_ptr())
        torch.cuda.synchronize()
        # counts only: the mixed-length lookup kernel then stores a round's counts together, without a pre-fill
        counts_only = torch.full((Q,), 9, dtype=torch.int32, device=dev)
        g.search_hits(chars_ptr, off_ptr, K, Q, 0, counts_only.data_ptr())
        torch.cuda.synchronize()
        if not torch.equal(counts_only, counts):
            print(f"COUNTS-ONLY MISMATCH n={n} k={seed_k} deep={deep_k} Q={Q} {desc} lookup_first={lookup!r} mixed_lookup={mixed!r}", flush=True)
            sys.exit(1)
        if g.search_hits_is_ordered(off_ptr != 0, K, Q):
            # results in search order, with the counts in that order: a permutation of the batch whose entries are the dense
            # results' (ranges of the k-mers with hits, every count)
            ok_ = torch.full((Q,), -1, dtype=torch.int32, device=dev)
            or_ = torch.full((Q * 2,), 9, dtype=torch.int64, device=dev)
            oc_ = torch.full((Q,), 9, dtype=torch.int32, device=dev)
            g.search_hits_in_order(chars_ptr, off_ptr, K, Q, ok_.data_ptr(), or_.data_ptr(), d_order_counts=oc_.data_ptr())
            torch.cuda.synchronize()
            ids = ok_.to(torch.int64)
            hit_ = counts[ids] != 0
            if not (int(torch.bincount(ids.clamp(0, Q - 1), minlength=Q).max().item()) == 1 and torch.equal(oc_, counts[ids])
                    and torch.equal(or_.view(Q, 2)[hit_], hits.view(Q, 2)[ids][hit_])
                    and bool((or_.view(Q, 2)[~hit_][:, 0] > or_.view(Q, 2)[~hit_][:, 1]).all())):
                print(f"SEARCH-ORDER MISMATCH n={n} k={seed_k} deep={deep_k} Q={Q} {desc} lookup_first={lookup!r} mixed_lookup={mixed!r}", flush=True)
                sys.exit(1)
        listed_ok = True
        if g.search_hits_is_ordered(off_ptr != 0, K, Q) and not fuzz_wide:
            # the list form of the same search, put in k-mer order, sized and located with nothing read back by the host
            cap = Q
            lk = torch.zeros(cap, dtype=torch.int32, device=dev)
            lr = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
            ln = torch.zeros(1, dtype=torch.int32, device=dev)
            g.search_hits_compact(chars_ptr, off_ptr, K, Q, lk.data_ptr(), lr.data_ptr(), cap, ln.data_ptr())
            g.sort_hits_on_device(lk.data_ptr(), lr.data_ptr(), cap, ln.data_ptr(), Q)
            torch.cuda.synchronize()
            m = int(ln.item())
            want = torch.nonzero(counts).flatten()
            listed_ok = (m == want.numel() and torch.equal(lk[:m].to(torch.int64), want)
                         and torch.equal(lr.view(cap, 2)[:m], hits.view(Q, 2)[want]))
            lengths = torch.where(lr.view(cap, 2)[:m, 0] <= lr.view(cap, 2)[:m, 1], lr.view(cap, 2)[:m, 1] - lr.view(cap, 2)[:m, 0] + 1,
                                  torch.zeros(m, dtype=torch.int64, device=dev))
            total_hits = int(lengths.sum().item())
            if listed_ok and m > 0 and ix.bwt_length < (1 << 32) and total_hits < 30_000_000:
                # the list's tail in one launch (lists beyond 2^18 entries: the three calls it replaces) against the list sorted
                # above, its offsets summed here and the positions of the general locate
                dense_tail = rng.random() < 0.5
                if dense_tail:
                    g.set_dense_sa(True)
                lk2 = torch.zeros(cap, dtype=torch.int32, device=dev)
                lr2 = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
                g.search_hits_compact(chars_ptr, off_ptr, K, Q, lk2.data_ptr(), lr2.data_ptr(), cap, ln.data_ptr())
                sk = torch.zeros(cap, dtype=torch.int32, device=dev)
                sr = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
                so = torch.zeros(cap + 1, dtype=torch.int64, device=dev)
                sp_ = torch.full((total_hits + 8,), -1, dtype=torch.int64, device=dev)
                g.list_locate_on_device(lk2.data_ptr(), lr2.data_ptr(), cap, ln.data_ptr(), Q, sk.data_ptr(), sr.data_ptr(), so.data_ptr(),
                                        total_hits, sp_.data_ptr())
                torch.cuda.synchronize()
                want_off = torch.zeros(m + 1, dtype=torch.int64, device=dev)
                torch.cumsum(lengths, 0, out=want_off[1:])
                ref_pos = torch.zeros(max(total_hits, 1), dtype=torch.int64, device=dev)
                if dense_tail:
                    g.set_dense_sa(False)
                g.locate(lr.data_ptr(), want_off.data_ptr(), m, total_hits, ref_pos.data_ptr())
                torch.cuda.synchronize()
                listed_ok = (torch.equal(sk[:m], lk[:m]) and torch.equal(sr[: 2 * m], lr[: 2 * m]) and torch.equal(so[: m + 1], want_off)
                             and torch.equal(sp_[:total_hits], ref_pos[:total_hits]) and bool((sp_[total_hits:] == -1).all()))
        g.set_wide(False)
        a, b = exact.view(Q, 2), hits.view(Q, 2)
        has = a[:, 0] <= a[:, 1]
        expect = torch.where(has, a[:, 1] - a[:, 0] + 1, torch.zeros_like(a[:, 0])).clamp(max=0xFFFFFFFF)
        ok = (torch.equal(a[has], b[has]) and bool((b[~has, 0] > b[~has, 1]).all())
              and torch.equal(counts.to(torch.int64) & 0xFFFFFFFF, expect))
        tag = f"n={n} k={seed_k} deep={deep_k} ratio={ratio} Q={Q} {desc} mis={mis} ordered={g.search_hits_is_ordered(off_ptr != 0, K, Q)} lookup_first={lookup!r} mixed_lookup={mixed!r}"
        if not listed_ok:
            print("LIST MISMATCH", tag, flush=True)
            sys.exit(1)
        if not ok:
            print("MISMATCH", tag, flush=True)
            sys.exit(1)
        if rounds % 4 == 0 and int(expect.sum()) < 50_000_000:
            # the locate pipeline on top of both: hit offsets (from ranges / from counts) and positions must agree
            scratch = torch.zeros(api.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
            off_a = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
            off_b = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
            total_a = g.hit_offsets(exact.data_ptr(), Q, off_a.data_ptr(), scratch.data_ptr())
            if ix.bwt_length < (1 << 32) and bool((expect < 0xFFFFFFFF).all()):
                total_b = g.hit_offsets_from_counts(counts.data_ptr(), Q, off_b.data_ptr(), scratch.data_ptr())
            else:
                total_b = g.hit_offsets(hits.data_ptr(), Q, off_b.data_ptr(), scratch.data_ptr())
            pos_a = torch.zeros(max(total_a, 1), dtype=torch.int64, device=dev)
            pos_b = torch.zeros(max(total_b, 1), dtype=torch.int64, device=dev)
            g.set_kernel(api.AWFM_GPU_KERNEL_GROUP2)  # side A: one LF step per block read
            g.locate(exact.data_ptr(), off_a.data_ptr(), Q, total_a, pos_a.data_ptr())
            torch.cuda.synchronize()
            g.set_kernel(api.AWFM_GPU_KERNEL_AUTO)
            g.set_wide(fuzz_wide)
            dense_sa = ix.bwt_length < (1 << 32) and rng.random() < 0.3  # side B through the full suffix array now and then
            if dense_sa:
                g.set_dense_sa(True)
            g.locate(hits.data_ptr(), off_b.data_ptr(), Q, total_b, pos_b.data_ptr())
            torch.cuda.synchronize()
            if dense_sa:
                g.set_dense_sa(False)
            g.set_wide(False)
            if not (total_a == total_b and torch.equal(off_a, off_b) and torch.equal(pos_a, pos_b)):
                print("LOCATE MISMATCH", tag, flush=True)
                sys.exit(1)
        if os.environ.get("FUZZ_VERBOSE"):
            print(tag, "hits", int(has.sum()), flush=True)
        rounds += 1
    g.destroy()
    ix.dealloc()
    del d_text
print(f"fuzz ok: {rounds} batches", flush=True)
