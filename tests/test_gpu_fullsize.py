"""Size-independent properties at (near) BASELINE sizes, where the oracle cannot be run over everything:
  * every planted k-mer is found, and every reported position really spells the k-mer in the text
    (the locate property of ref test/parallelSearch/parallelSearchTest.c:105-214, checked on the device);
  * hit lists are strictly inside [0, n-K] and, per query, pairwise distinct;
  * counts equal range lengths; an oracle-checked sample of the same batch agrees bit for bit.
Sizes: AWFM_TEST_TEXT_LEN / AWFM_TEST_QUERIES (defaults 400 Mbp / 20 M; bench.py runs the 3.1 Gbp / 100 M
case with the same parity gate)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_planted_kmers_are_located_where_they_were_taken(oracle, awfm, require_gpu):
    import torch
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    n = int(os.environ.get("AWFM_TEST_TEXT_LEN", 400_000_000))
    Q = int(os.environ.get("AWFM_TEST_QUERIES", 20_000_000))
    K = 21
    dev = torch.device("cuda")
    d_text = torch.empty(n, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 2, 0, None) == 1
    ix = awfm.gpu_create_index(d_text.data_ptr(), awfm.AwFmAlphabetDna, 8, 12, on_device_length=n)
    g = awfm.GpuIndex(ix, acquire=True)
    d_chars = torch.empty(Q * K, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthPlantedQueries(d_chars.data_ptr(), 0, Q, K, 103, d_text.data_ptr(), n, None) == 1
    d_ranges = torch.empty(Q * 2, dtype=torch.int64, device=dev)
    d_counts = torch.empty(Q, dtype=torch.int32, device=dev)
    d_off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
    d_scratch = torch.empty(awfm.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
    g.search(d_chars.data_ptr(), 0, K, Q, d_ranges.data_ptr(), d_counts.data_ptr())
    total = g.hit_offsets(d_ranges.data_ptr(), Q, d_off.data_ptr(), d_scratch.data_ptr())
    d_pos = torch.empty(total, dtype=torch.int64, device=dev)
    g.locate(d_ranges.data_ptr(), d_off.data_ptr(), Q, total, d_pos.data_ptr())
    torch.cuda.synchronize()
    r = d_ranges.view(Q, 2)
    lens = torch.where(r[:, 0] <= r[:, 1], r[:, 1] - r[:, 0] + 1, torch.zeros_like(r[:, 0]))
    assert int(lens.min()) >= 1, "a planted k-mer was not found"
    assert torch.equal(lens.to(torch.int32), d_counts) and int(lens.sum()) == total
    assert torch.equal(torch.cumsum(lens, 0), d_off[1:]) and int(d_off[0]) == 0
    assert int(d_pos.min()) >= 0 and int(d_pos.max()) <= n - K
    # every hit spells its k-mer: compare text[pos + c] with the query's character c, one column at a time
    owner = torch.repeat_interleave(torch.arange(Q, device=dev), lens)
    q2d = d_chars.view(Q, K)
    for c in range(K):
        assert torch.equal(d_text[d_pos + c], q2d[owner, c]), f"hit does not match the k-mer at character {c}"
    # the seeded offset each k-mer was copied from is among its hits
    from avxwindowfmindex_amd import synth
    sample = 200_000
    planted = torch.from_numpy(synth.planted_offsets(103, sample, K, n).astype(np.int64)).to(dev)
    first_hit = d_off[:sample]
    found = torch.zeros(sample, dtype=torch.bool, device=dev)
    maxc = int(lens[:sample].max())
    for h in range(maxc):
        valid = lens[:sample] > h
        idx = torch.where(valid, first_hit + h, torch.zeros_like(first_hit))
        found |= valid & (d_pos[idx] == planted)
    assert bool(found.all())
    # hits of one query are pairwise distinct (BWT order gives distinct suffixes)
    multi = torch.nonzero(lens[:sample] > 1).flatten()[:2000].tolist()
    pos_cpu = d_pos[: int(d_off[sample])].cpu().numpy()
    off_cpu = d_off[: sample + 1].cpu().numpy()
    for q in multi:
        hits = pos_cpu[off_cpu[q]:off_cpu[q + 1]]
        assert len(set(hits.tolist())) == len(hits)
    # oracle on a sample of the same batch, bit for bit
    m = 300_000
    oi = oracle.Index.wrap(oracle.DNA, 8, 12, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    chars = d_chars[: m * K].cpu().numpy()
    offsets = np.arange(m + 1, dtype=np.uint64) * np.uint64(K)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=os.cpu_count() or 1)
    ho, pos, _ = oi.batch_locate(sp, ep, threads=os.cpu_count() or 1)
    gr = d_ranges[: 2 * m].cpu().numpy().view(np.uint64).reshape(m, 2)
    assert np.array_equal(gr[:, 0], sp) and np.array_equal(gr[:, 1], ep)
    assert np.array_equal(d_off[: m + 1].cpu().numpy().view(np.uint64), ho)
    assert np.array_equal(d_pos[: int(ho[-1])].cpu().numpy().view(np.uint64), pos)
    g.destroy()
    ix.dealloc()


def test_random_kmers_counts_are_consistent_between_count_and_locate(awfm, require_gpu):
    """idempotence / agreement: the same batch through the count path and the locate path, host-buffer API"""
    from avxwindowfmindex_amd import synth
    txt = synth.text(2, 3_000_000)
    ix = awfm.gpu_create_index(txt, awfm.AwFmAlphabetDna, 8, 10)
    g = awfm.GpuIndex(ix, acquire=True)
    q = np.concatenate([synth.random_queries(102, 300000, 14), synth.planted_queries(103, 300000, 14, txt)])
    chars, _ = synth.fixed_csr(q)
    ranges, counts = g.count_host(chars, None, fixed_length=14)
    r2, ho, pos = g.locate_host(chars, None, fixed_length=14)
    assert np.array_equal(ranges, r2) and np.array_equal(np.diff(ho).astype(np.uint32), counts)
    raw = txt.tobytes()
    for j in list(range(0, 600000, 997)):
        k = q[j].tobytes()
        for p in pos[int(ho[j]):int(ho[j + 1])].tolist():
            assert raw[p:p + 14] == k
    assert counts[300000:].min() >= 1
    g.destroy()
    ix.dealloc()


def test_batches_beyond_4_gib_of_query_characters(oracle, awfm, require_gpu):
    """byte offsets into the query buffer exceed 2^32 (fixed-length and CSR): the tail of the batch is
    compared with the oracle"""
    import torch
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    n, K = 50_000_000, 21
    Q = (1 << 32) // K + 3_000_000  # ~207.5 M k-mers, 4.36 GB of characters
    dev = torch.device("cuda")
    d_text = torch.empty(n, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 6, 0, None) == 1
    ix = awfm.gpu_create_index(d_text.data_ptr(), awfm.AwFmAlphabetDna, 8, 10, on_device_length=n)
    g = awfm.GpuIndex(ix, acquire=True)
    d_chars = torch.empty(Q * K, dtype=torch.uint8, device=dev)
    half = Q // 2
    assert L.awfmGpuSynthRandomQueries(d_chars.data_ptr(), 0, half, K, 106, 0, None) == 1
    assert L.awfmGpuSynthPlantedQueries(d_chars.data_ptr() + half * K, half, Q - half, K, 107, d_text.data_ptr(), n, None) == 1
    d_ranges = torch.empty(Q * 2, dtype=torch.int64, device=dev)
    g.search(d_chars.data_ptr(), 0, K, Q, d_ranges.data_ptr(), 0)
    torch.cuda.synchronize()
    fixed = d_ranges.clone()
    d_offsets = torch.arange(Q + 1, dtype=torch.int64, device=dev) * K
    d_ranges.zero_()
    g.search(d_chars.data_ptr(), d_offsets.data_ptr(), 0, Q, d_ranges.data_ptr(), 0)
    torch.cuda.synchronize()
    assert torch.equal(fixed, d_ranges), "CSR and fixed-length addressing disagree"
    m = 200_000  # the last m k-mers: their bytes start beyond 4 GiB
    assert (Q - m) * K > (1 << 32)
    oi = oracle.Index.wrap(oracle.DNA, 8, 10, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    chars = d_chars[(Q - m) * K:].cpu().numpy()
    sp, ep, _, _ = oi.batch_search(chars, np.arange(m + 1, dtype=np.uint64) * np.uint64(K), threads=8)
    tail = d_ranges[2 * (Q - m):].cpu().numpy().view(np.uint64).reshape(m, 2)
    assert np.array_equal(tail[:, 0], sp) and np.array_equal(tail[:, 1], ep)
    assert int((tail[:, 0] <= tail[:, 1]).sum()) == m  # the second half is planted
    g.destroy()
    ix.dealloc()


def test_hits_only_search_takes_the_ordered_path_by_itself_and_agrees_with_the_general_kernel(oracle, awfm, require_gpu):
    """automatic mode: 8.5 M k-mers against a 280 Mbp index cross both thresholds (2^23 queries, 2^28 positions);
    every one of them is compared with awfmGpuSearch on the device, a sample with the oracle"""
    import torch
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    n, K, Q = 280_000_000, 21, 8_500_000
    dev = torch.device("cuda")
    d_text = torch.empty(n, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 8, 0, None) == 1
    ix = awfm.gpu_create_index(d_text.data_ptr(), awfm.AwFmAlphabetDna, 8, 11, on_device_length=n)
    g = awfm.GpuIndex(ix, acquire=True)
    d_chars = torch.empty(Q * K + 8, dtype=torch.uint8, device=dev)
    half = Q // 2
    assert L.awfmGpuSynthRandomQueries(d_chars.data_ptr(), 0, half, K, 108, 0, None) == 1
    assert L.awfmGpuSynthPlantedQueries(d_chars.data_ptr() + half * K, half, Q - half, K, 109, d_text.data_ptr(), n, None) == 1
    del d_text
    assert g.search_hits_is_ordered(False, K, Q) and not g.search_hits_is_ordered(False, K, 1000)
    exact = torch.zeros(Q * 2, dtype=torch.int64, device=dev)
    hits = torch.full((Q * 2,), 5, dtype=torch.int64, device=dev)
    counts = torch.full((Q,), 5, dtype=torch.int32, device=dev)
    g.search(d_chars.data_ptr(), 0, K, Q, exact.data_ptr(), 0)
    g.search_hits(d_chars.data_ptr(), 0, K, Q, hits.data_ptr(), counts.data_ptr())
    torch.cuda.synchronize()
    a, b = exact.view(Q, 2), hits.view(Q, 2)
    has = a[:, 0] <= a[:, 1]
    assert int(has.sum()) >= Q - half
    assert torch.equal(a[has], b[has]), "ranges of k-mers with hits differ between the two kernels"
    assert bool((b[~has, 0] > b[~has, 1]).all())
    assert torch.equal(counts.to(torch.int64), torch.where(has, a[:, 1] - a[:, 0] + 1, torch.zeros_like(a[:, 0])))
    m = 100_000  # the last m k-mers (planted) and the first m (random) against the oracle
    oi = oracle.Index.wrap(oracle.DNA, 8, 11, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    for lo in (0, Q - m):
        chars = d_chars[lo * K:(lo + m) * K].cpu().numpy()
        sp, ep, cnt, _ = oi.batch_search(chars, np.arange(m + 1, dtype=np.uint64) * np.uint64(K), threads=8)
        got = hits[2 * lo:2 * (lo + m)].cpu().numpy().view(np.uint64).reshape(m, 2)
        hit = cnt > 0
        assert np.array_equal(got[hit, 0], sp[hit]) and np.array_equal(got[hit, 1], ep[hit])
        assert np.array_equal(counts[lo:lo + m].cpu().numpy().view(np.uint32), cnt)
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("alphabet", ["dna", "amino"])
def test_an_index_beyond_2_pow_32_positions_built_searched_and_located_on_the_gpu(oracle, awfm, require_gpu, alphabet):
    """bwtLength > 2^32 for real (ref src/AwFmIndex.h:55-65, :88-91 and src/AwFmSuffixArray.c:12-18 are 64-bit
    throughout): the builder's 64-bit suffix sort, 33-bit sampled SA values, two nucleotide superblocks, the 64-bit
    search / walk kernels and the pair image's global superblock table, with nothing forced by a knob.  Planted
    24-mers -- a share of them taken beyond position 2^32 -- must come back at their planting offsets, every hit must
    spell its k-mer, and an oracle-checked sample of the batch (planted and random k-mers) must agree bit for bit.
    The amino case runs the same through the 128-byte amino blocks with their 2^16-position superblocks and 64-bit
    bases (14-mers over a seed table of depth 4)."""
    import time
    import torch
    from avxwindowfmindex_amd import _lib, synth
    L = _lib.lib()
    n = int(os.environ.get("AWFM_TEST_WIDE_TEXT_LEN", (1 << 32) + 100_000_000))
    Q = int(os.environ.get("AWFM_TEST_WIDE_QUERIES", 10_000_000))
    amino = alphabet == "amino"
    K, seed_k = (14, 4) if amino else (24, 12)
    kind = awfm.AwFmAlphabetAmino if amino else awfm.AwFmAlphabetDna
    dev = torch.device("cuda")
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * n:
        pytest.skip(f"needs about {40 * n >> 30} GiB of HBM for the 64-bit suffix sort, {free >> 30} GiB free")
    d_text = torch.empty(n, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 7, 1 if amino else 0, None) == 1
    t0 = time.time()
    ix = awfm.gpu_create_index(d_text.data_ptr(), kind, 8, seed_k, on_device_length=n)
    print(f"\n[wide build] {n} {alphabet} characters in {time.time() - t0:.1f} s, SA width {ix.sa_width} bits")
    assert ix.bwt_length == n + 1 and (n < (1 << 32) or ix.sa_width == 33)
    g = awfm.GpuIndex(ix, acquire=True)
    assert g.has_pair_image == (not amino)
    half = Q // 2
    d_chars = torch.empty(Q * K, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthPlantedQueries(d_chars.data_ptr(), 0, half, K, 203, d_text.data_ptr(), n, None) == 1
    assert L.awfmGpuSynthRandomQueries(d_chars.data_ptr() + half * K, 0, Q - half, K, 204, 1 if amino else 0, None) == 1
    d_ranges = torch.empty(Q * 2, dtype=torch.int64, device=dev)
    d_hits = torch.empty(Q * 2, dtype=torch.int64, device=dev)
    d_counts = torch.empty(Q, dtype=torch.int32, device=dev)
    d_counts2 = torch.empty(Q, dtype=torch.int32, device=dev)
    d_off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
    d_scratch = torch.empty(awfm.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
    g.search(d_chars.data_ptr(), 0, K, Q, d_ranges.data_ptr(), d_counts.data_ptr())  # exact ranges, general kernel
    g.set_ordered(1)
    g.search_hits(d_chars.data_ptr(), 0, K, Q, d_hits.data_ptr(), d_counts2.data_ptr())  # seed order + pair steps
    torch.cuda.synchronize()
    assert torch.equal(d_counts, d_counts2)
    hit = d_counts > 0
    assert torch.equal(d_ranges.view(Q, 2)[hit], d_hits.view(Q, 2)[hit])
    assert bool(hit[:half].all()), "a planted k-mer was not found"
    assert int(d_ranges.view(Q, 2)[hit].max()) > (1 << 32) or n < (1 << 32)
    total = g.hit_offsets(d_hits.data_ptr(), Q, d_off.data_ptr(), d_scratch.data_ptr())
    d_pos = torch.empty(total, dtype=torch.int64, device=dev)
    g.locate(d_hits.data_ptr(), d_off.data_ptr(), Q, total, d_pos.data_ptr())
    torch.cuda.synchronize()
    lens = d_counts.to(torch.int64)
    assert int(lens.sum()) == total and int(d_pos.min()) >= 0 and int(d_pos.max()) <= n - K
    owner = torch.repeat_interleave(torch.arange(Q, device=dev), lens)
    q2d = d_chars.view(Q, K)
    for c in range(K):
        assert torch.equal(d_text[d_pos + c], q2d[owner, c]), f"hit does not match the k-mer at character {c}"
    planted = torch.from_numpy(synth.planted_offsets(203, half, K, n).astype(np.int64)).to(dev)
    beyond = planted >= (1 << 32)
    assert n < (1 << 32) or int(beyond.sum()) > half // 100
    single = lens[:half] == 1
    assert float(single.float().mean()) > 0.99
    assert torch.equal(d_pos[d_off[:half][single]], planted[single])
    assert n < (1 << 32) or bool((beyond & single).any())
    if not amino and n >= (1 << 32):
        # round 6: the FAST path of an image beyond 2^32 positions, nothing forced -- the deeper table in its packed 8-byte form
        # with next-step bits, the full suffix array in 40-bit entries, and, for a batch of random 21-mers in the list form
        # (bench.py's timed step), lookupSearchKernel's 64-bit instantiation: it must be the kernel that ran, and its list,
        # offsets and positions must be those of the exact general search and of the LF walk
        assert g.is_wide and g.deep_seed_k == 16 and g.has_dense_sa and "next-step bits" in g.describe(), g.describe()
        Kr, Qr = 21, Q
        d_rand = torch.empty(Qr * Kr + 8, dtype=torch.uint8, device=dev)
        assert L.awfmGpuSynthRandomQueries(d_rand.data_ptr(), 0, Qr, Kr, 205, 0, None) == 1
        cap = Qr // 64
        lk = torch.empty(cap, dtype=torch.int32, device=dev)
        lr = torch.empty(cap * 2, dtype=torch.int64, device=dev)
        sk = torch.empty(cap, dtype=torch.int32, device=dev)
        sr = torch.empty(cap * 2, dtype=torch.int64, device=dev)
        lo_ = torch.empty(cap + 1, dtype=torch.int64, device=dev)
        num = torch.zeros(1, dtype=torch.int32, device=dev)
        lpos = torch.empty(Qr // 16, dtype=torch.int64, device=dev)
        st = torch.cuda.Stream()
        for _ in range(3):  # (the third call is predicted: the lookup kernel alone)
            g.search_hits_compact(d_rand.data_ptr(), 0, Kr, Qr, lk.data_ptr(), lr.data_ptr(), cap, num.data_ptr(), stream=st.cuda_stream)
            g.list_locate_on_device(lk.data_ptr(), lr.data_ptr(), cap, num.data_ptr(), Qr, sk.data_ptr(), sr.data_ptr(), lo_.data_ptr(),
                                    lpos.numel(), lpos.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        assert g.last_ordered_kernel_is_lookup(), "the lookup kernel did not take a batch of random 21-mers on an image beyond 2^32 positions"
        assert g.last_lookup_front() == 1
        m_ = int(num.item())
        ex = torch.empty(Qr * 2, dtype=torch.int64, device=dev)
        g.search(d_rand.data_ptr(), 0, Kr, Qr, ex.data_ptr(), 0)  # exact ranges
        torch.cuda.synchronize()
        ex2 = ex.view(Qr, 2)
        want = torch.nonzero(ex2[:, 0] <= ex2[:, 1]).flatten()
        assert m_ == want.numel() and m_ > 1000 and torch.equal(sk[:m_].to(torch.int64), want)
        assert torch.equal(sr.view(cap, 2)[:m_], ex2[want])
        tot = int(lo_[cap].item())
        assert tot == int((ex2[want, 1] - ex2[want, 0] + 1).sum()) and tot <= lpos.numel()
        g.set_dense_sa(False)  # the same list through the LF walk and the sampled array (the reference's backtrace)
        walked = torch.empty(tot, dtype=torch.int64, device=dev)
        g.locate(sr.data_ptr(), lo_.data_ptr(), m_, tot, walked.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(walked, lpos[:tot]), "the 40-bit full suffix array and the LF walk disagree"
        own = torch.repeat_interleave(want, ex2[want, 1] - ex2[want, 0] + 1)
        r2d = d_rand[: Qr * Kr].view(Qr, Kr)
        for c in range(Kr):
            assert torch.equal(d_text[walked + c], r2d[own, c]), f"a random k-mer's hit does not spell it at character {c}"
        # ... and the array once more, this time put together from the sampled one (what an index read from a file gets: capped
        # walks, parked ones completed by pointer jumping, 64-bit entries packed to 40 bits): the same positions again
        t1 = time.time()
        g.set_dense_sa(True)
        torch.cuda.synchronize()
        print(f"[wide full suffix array from the sampled one] {time.time() - t1:.1f} s")
        assert g.has_dense_sa
        lpos.fill_(-1)
        g.list_locate_on_device(lk.data_ptr(), lr.data_ptr(), cap, num.data_ptr(), Qr, sk.data_ptr(), sr.data_ptr(), lo_.data_ptr(),
                                lpos.numel(), lpos.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(walked, lpos[:tot]), "the full suffix array built from the sampled one gives other positions"
        del d_rand, ex, walked, lpos
    # oracle over the downloaded (reference-layout) arrays on a sample from both halves of the batch
    oi = oracle.Index.wrap(oracle.AMINO if amino else oracle.DNA, 8, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(),
                           ix.seed_table(), ix.packed_sa())
    m = 100_000
    for first in (0, half):
        chars = d_chars[first * K:(first + m) * K].cpu().numpy()
        offsets = np.arange(m + 1, dtype=np.uint64) * np.uint64(K)
        sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=os.cpu_count() or 1)
        ho, pos, _ = oi.batch_locate(sp, ep, threads=os.cpu_count() or 1)
        gr = d_ranges[2 * first: 2 * (first + m)].cpu().numpy().view(np.uint64).reshape(m, 2)
        assert np.array_equal(gr[:, 0], sp) and np.array_equal(gr[:, 1], ep), "exact ranges differ from the oracle's"
        base = int(d_off[first])
        assert np.array_equal(d_off[first: first + m + 1].cpu().numpy().view(np.uint64) - np.uint64(base), ho)
        assert np.array_equal(d_pos[base: base + int(ho[-1])].cpu().numpy().view(np.uint64), pos)
    g.destroy()
    ix.dealloc()


def test_full_suffix_array_of_a_genome_shaped_index_at_full_size(awfm, require_gpu, monkeypatch):
    """The full suffix array of a LOADED genome-shaped 3.1 Gbp index (24 runs of N of 10^5..10^7 characters: LF walks that never
    meet a sample until a run ends) -- the automatic construction (capped walks, the parked ones completed from each other by
    pointer jumping) and, since round 5, the one that is asked for -- against the array the GPU builder sorted: an image
    acquired from the host arrays alone must have its array within seconds (it took 566 s before round 4's last day), and
    10^6 k-mers drawn from the text, the 21-mers right behind every run and a 21-mer of N (10^8 hits, all inside the runs) must
    come back at the builder's positions; the LF walk itself (no array) agrees on the 10^6."""
    import time
    import torch
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    n = int(os.environ.get("AWFM_TEST_GENOME_LEN", 3_100_000_000))
    Q, K = 1_000_000, 21
    dev = torch.device("cuda")
    d_text = torch.empty(n, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthGenomeText(d_text.data_ptr(), n, 2, None) == 1
    ix = awfm.gpu_create_index(d_text.data_ptr(), awfm.AwFmAlphabetDna, 8, 12, on_device_length=n)
    g0 = awfm.GpuIndex(ix, acquire=True)
    assert g0.has_dense_sa and g0.dense_sa_build_s < 0.5, "the image of an index built on the GPU takes the builder's own array"
    d_q = torch.empty(Q * K, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthPlantedQueriesUnique(d_q.data_ptr(), 0, Q, K, 7, d_text.data_ptr(), n, 2, None, None) == 1
    ends = []  # the first character behind every run of N (in pieces: torch.nonzero indexes with 32 bits)
    piece = 1 << 30
    for b in range(0, n - 1, piece):
        e = min(n - 1, b + piece)
        is_n = d_text[b:e + 1] == ord("n")
        ends += (torch.nonzero(is_n[:-1] & ~is_n[1:]).flatten() + (b + 1)).tolist()
        del is_n
    behind = [d_text[int(e): int(e) + K].cpu().numpy().tobytes() for e in ends if int(e) + K <= n]
    behind = [k for k in behind if b"n" not in k]
    assert len(behind) >= 20, "the text has 24 runs of N"
    chars = np.concatenate([d_q.cpu().numpy(), np.frombuffer(b"".join(behind), np.uint8), np.frombuffer(b"n" * K, np.uint8)])
    total = Q + len(behind) + 1
    d_chars = torch.from_numpy(chars).to(dev)
    del d_text, d_q
    torch.cuda.empty_cache()

    def locate(g, count):
        d_ranges = torch.empty(count * 2, dtype=torch.int64, device=dev)
        g.search(d_chars.data_ptr(), 0, K, count, d_ranges.data_ptr(), 0)
        d_off = torch.empty(count + 1, dtype=torch.int64, device=dev)
        d_scratch = torch.empty(awfm.GpuIndex.scan_scratch_bytes(count), dtype=torch.uint8, device=dev)
        hits = g.hit_offsets(d_ranges.data_ptr(), count, d_off.data_ptr(), d_scratch.data_ptr())
        d_pos = torch.empty(max(hits, 1), dtype=torch.int64, device=dev)
        g.locate(d_ranges.data_ptr(), d_off.data_ptr(), count, hits, d_pos.data_ptr())
        torch.cuda.synchronize()
        return d_off, d_pos

    off0, pos0 = locate(g0, total)
    assert int(off0[-1] - off0[-2]) > n // 100, "the k-mer of N has its hits inside the runs"
    # the image goes; the next one comes from the host arrays alone, as for an index read from a file
    g0.handle = None
    L.awfmGpuIndexRelease(ix.ptr)
    t0 = time.perf_counter()
    g1 = awfm.GpuIndex(ix, acquire=True)
    acquire_s = time.perf_counter() - t0
    # (seconds, not the 566 s of walking every position to the end: 0.9 s of kernels; a fresh box's first large hipMalloc calls
    # have taken 5 s by themselves -- round 6's first run on the pool: 5.5 s --, hence the margin)
    assert g1.has_dense_sa and g1.dense_sa_build_s < 15.0, (g1.has_dense_sa, g1.dense_sa_build_s, acquire_s)
    off1, pos1 = locate(g1, total)
    assert torch.equal(off0, off1) and torch.equal(pos0, pos1), "the automatic full suffix array gives other positions than the builder's"
    # no array: the LF walk (the 10^6 k-mers from the unique sequence; a walk from behind a run is as long as the run)
    g1.set_dense_sa(False)
    off2, pos2 = locate(g1, Q)
    assert torch.equal(off0[: Q + 1], off2) and torch.equal(pos0[: int(off0[Q])], pos2), "the LF walk gives other positions"
    # the array that is asked for: capped walks and pointer jumping as well
    t0 = time.perf_counter()
    g1.set_dense_sa(True)
    torch.cuda.synchronize()
    asked_s = time.perf_counter() - t0
    assert g1.has_dense_sa and asked_s < 15.0, asked_s
    off3, pos3 = locate(g1, total)
    assert torch.equal(off0, off3) and torch.equal(pos0, pos3), "the full suffix array that was asked for gives other positions"
    g1.handle = None
    ix.dealloc()


def test_an_image_created_when_memory_is_short_says_what_it_dropped_and_answers_the_same(oracle, awfm, require_gpu):
    """Every device-only accelerator is dropped silently when the device is short of memory (the deeper table wants three
    times its size free, the full suffix array four times): with all but 3 GB of the device taken, a 3*10^8-position image
    must come up without either, SAY so (awfmGpuIndexDescribe), and search and locate 2*10^6 k-mers through the general
    kernel and the LF walk with the results the same image gives once the memory is back and both are built -- and the
    oracle's on a sample.  (Round 5's verdict: the low-memory path was visible in bench.py's line only.)"""
    import torch
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    n, Q, K = 300_000_000, 2_000_000, 21
    dev = torch.device("cuda")
    d_text = torch.empty(n, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 5, 0, None) == 1
    ix = awfm.gpu_create_index(d_text.data_ptr(), awfm.AwFmAlphabetDna, 8, 12, on_device_length=n)
    L.awfmGpuIndexRelease(ix.ptr)  # the image the builder left (it has its accelerators): this test makes its own
    d_chars = torch.empty(Q * K, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthPlantedQueries(d_chars.data_ptr(), 0, Q // 2, K, 31, d_text.data_ptr(), n, None) == 1
    assert L.awfmGpuSynthRandomQueries(d_chars.data_ptr() + (Q // 2) * K, 0, Q - Q // 2, K, 32, 0, None) == 1
    torch.cuda.synchronize()
    del d_text
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(free - (3 << 30), dtype=torch.uint8, device=dev)
    g = awfm.GpuIndex(ix)
    said = g.describe()
    assert g.deep_seed_k == 0 and not g.has_dense_sa, said
    assert "deeper table: depth 14 not built" in said and "full suffix array: not built" in said, said

    def run():
        ranges = torch.full((Q * 2,), 7, dtype=torch.int64, device=dev)
        counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
        off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
        scratch = torch.empty(awfm.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
        g.search_hits(d_chars.data_ptr(), 0, K, Q, ranges.data_ptr(), counts.data_ptr())
        total = g.hit_offsets_from_counts(counts.data_ptr(), Q, off.data_ptr(), scratch.data_ptr())
        pos = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
        g.locate(ranges.data_ptr(), off.data_ptr(), Q, total, pos.data_ptr())
        torch.cuda.synchronize()
        return ranges.view(Q, 2), counts, off, pos[:total]

    r0, c0, o0, p0 = run()
    assert int(c0[: Q // 2].min()) >= 1, "a k-mer drawn from the text was not found"
    del hog
    torch.cuda.empty_cache()
    g.set_deep_seed(14)
    g.set_dense_sa(True)
    assert g.deep_seed_k == 14 and g.has_dense_sa
    r1, c1, o1, p1 = run()
    has = c0 > 0
    assert torch.equal(c0, c1) and torch.equal(o0, o1) and torch.equal(p0, p1) and torch.equal(r0[has], r1[has])
    m = 100_000  # the oracle over the same arrays: the first planted and the first random k-mers
    oi = oracle.Index.wrap(oracle.DNA, 8, 12, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    for first in (0, Q // 2):
        chars = d_chars[first * K: (first + m) * K].cpu().numpy()
        sp, ep, cnt, _ = oi.batch_search(chars, np.arange(m + 1, dtype=np.uint64) * np.uint64(K), threads=os.cpu_count() or 1)
        ho, pos, _ = oi.batch_locate(sp, ep, threads=os.cpu_count() or 1)
        assert np.array_equal(c0[first: first + m].cpu().numpy().view(np.uint32), cnt)
        hit = cnt > 0
        rr = r0[first: first + m].cpu().numpy().view(np.uint64)
        assert np.array_equal(rr[hit, 0], sp[hit]) and np.array_equal(rr[hit, 1], ep[hit])
        a, b = int(o0[first]), int(o0[first + m])
        assert np.array_equal(p0[a:b].cpu().numpy().view(np.uint64), pos)
    g.destroy()
    ix.dealloc()
