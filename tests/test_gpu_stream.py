"""The chunked host-buffer pipeline with bit-packed k-mers (include/awfm_gpu.h, csrc/awfm_gpu_stream.hip) against the
CPU oracle: counts and flat position lists of every chunk must be exactly what awFmParallelSearchCount/Locate report
per k-mer (ref src/AwFmParallelSearch.c:95-365: count = range length, positions in BWT order), whatever the chunk
size, the search path a chunk takes, and whether the input is pageable or page-locked."""
import ctypes as C

import numpy as np
import pytest

from avxwindowfmindex_amd import synth

pytestmark = pytest.mark.gpu


def _oracle_answers(O, oalpha, ix, ratio, seed_k, kmers):
    oi = O.Index.wrap(oalpha, ratio, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    chars, offsets = synth.fixed_csr(kmers)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    _, pos, _ = oi.batch_locate(sp, ep)
    return cnt, pos


@pytest.mark.parametrize("ordered", [0, 1])
@pytest.mark.parametrize("chunk", [0, 777, 4096])
def test_packed_dna_stream_matches_oracle(oracle, awfm, require_gpu, wide, chunk, ordered):
    n, K = 300_000, 21
    txt = synth.text(31, n).copy()
    txt[1000:1003] = ord("n")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    g = awfm.GpuIndex(ix)
    g.set_ordered(ordered)  # 1: every chunk takes the seed-order path, 0: the general kernel
    kmers = np.concatenate([synth.random_queries(32, 5000, K), synth.planted_queries(33, 5001, K, synth.text(31, n))])
    cnt, pos = _oracle_answers(oracle, oracle.DNA, ix, 8, 8, kmers)
    packed = awfm.pack_kmers(kmers)
    counts, positions = g.stream(packed, K, locate=True, chunk=chunk)
    assert np.array_equal(counts, cnt), "counts differ"
    assert np.array_equal(positions, pos), "positions differ (BWT order per k-mer, k-mers in batch order)"
    counts_only, none = g.stream(packed, K, locate=False, chunk=chunk)
    assert none is None and np.array_equal(counts_only, cnt)
    # whole-batch convenience calls
    assert np.array_equal(g.count_packed_host(packed, K), cnt)
    c2, p2 = g.locate_packed_host(packed, K)
    assert np.array_equal(c2, cnt) and np.array_equal(p2, pos)
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("K,seed_k", [(10, 3), (12, 2), (1, 2)])
def test_packed_amino_stream_matches_oracle(oracle, awfm, require_gpu, K, seed_k):
    n = 120_000
    txt = synth.text(41, n, synth.AMINO_ALPHABET)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetAmino, 8, seed_k)
    g = awfm.GpuIndex(ix)
    kmers = np.concatenate([synth.random_queries(42, 3000, K, synth.AMINO_ALPHABET), synth.planted_queries(43, 3000, K, txt)])
    cnt, pos = _oracle_answers(oracle, oracle.AMINO, ix, 8, seed_k, kmers)
    packed = awfm.pack_kmers(kmers, awfm.AwFmAlphabetAmino)
    counts, positions = g.stream(packed, K, locate=True, chunk=1500)
    assert np.array_equal(counts, cnt) and np.array_equal(positions, pos)
    g.destroy()
    ix.dealloc()


def test_ascii_stream_with_ambiguity_characters(oracle, awfm, require_gpu):
    """the ASCII form of the pipeline takes everything the ASCII API takes: mixed case, ambiguity letters"""
    n, K = 200_000, 16
    txt = synth.text(51, n).copy()
    txt[500:520] = ord("n")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 5, 6)
    g = awfm.GpuIndex(ix)
    kmers = np.concatenate([synth.random_queries(52, 4000, K), synth.planted_queries(53, 4000, K, synth.text(51, n))]).copy()
    rng = np.random.default_rng(5)
    kmers[rng.random(kmers.shape) < 0.01] = ord("x")
    up = rng.random(kmers.shape) < 0.3
    kmers[up] &= 0xDF
    kmers[-1] = np.frombuffer(b"nnnnnnnnnnnnnnnn", np.uint8)  # matches the sanitised run in the text
    cnt, pos = _oracle_answers(oracle, oracle.DNA, ix, 5, 6, kmers)
    assert cnt[-1] == 5
    counts, positions = g.stream(kmers.reshape(-1), K, locate=True, chunk=3000, packed=False)
    assert np.array_equal(counts, cnt) and np.array_equal(positions, pos)
    g.destroy()
    ix.dealloc()


def test_page_locked_input_sink_order_and_abort(oracle, awfm, require_gpu):
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    n, K, Q = 100_000, 21, 10_000
    txt = synth.text(61, n)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    g = awfm.GpuIndex(ix)
    kmers = synth.planted_queries(62, Q, K, txt)
    cnt, pos = _oracle_answers(oracle, oracle.DNA, ix, 8, 8, kmers)
    packed = awfm.pack_kmers(kmers)
    address = L.awfmGpuHostAlloc(Q * 8)  # the DMA engine reads the batch where it lies
    assert address
    C.memmove(address, packed.ctypes.data, Q * 8)
    seen = []

    def sink(user, first, m, counts, positions, total):
        c = np.ctypeslib.as_array(counts, shape=(m,)).copy()
        p = np.ctypeslib.as_array(positions, shape=(total,)).copy() if total else np.zeros(0, np.uint64)
        seen.append((first, m, c, p))
        return 0

    g.stream((address, Q), K, locate=True, chunk=1024, sink=sink)
    assert [s[0] for s in seen] == list(range(0, Q, 1024)), "chunks arrive in batch order"
    assert sum(s[1] for s in seen) == Q
    assert np.array_equal(np.concatenate([s[2] for s in seen]), cnt)
    assert np.array_equal(np.concatenate([s[3] for s in seen]), pos)
    # a sink that returns non-zero stops the batch and the call reports it
    calls = []

    def stop(user, first, m, counts, positions, total):
        calls.append(first)
        return 1

    with pytest.raises(awfm.AwFmError):
        g.stream((address, Q), K, locate=True, chunk=1024, sink=stop)
    assert calls == [0]
    # the pipeline is usable again afterwards
    counts, positions = g.stream(packed, K, locate=True, chunk=4000)
    assert np.array_equal(counts, cnt) and np.array_equal(positions, pos)
    L.awfmGpuHostFree(address)
    g.destroy()
    ix.dealloc()


def test_device_pack_and_unpack_round_trip(awfm, require_gpu):
    import torch
    n, K, Q = 50_000, 21, 20_000
    txt = synth.text(71, n)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 6)
    g = awfm.GpuIndex(ix)
    kmers = synth.random_queries(72, Q, K).copy()
    d_chars = torch.from_numpy(kmers.reshape(-1)).cuda()
    d_packed = torch.empty(Q, dtype=torch.int64, device="cuda")
    assert g.pack_device(d_chars.data_ptr(), K, Q, d_packed.data_ptr()) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_packed.cpu().numpy().view(np.uint64), awfm.pack_kmers(kmers))
    d_back = torch.empty(Q * K, dtype=torch.uint8, device="cuda")
    g.unpack_device(d_packed.data_ptr(), K, Q, d_back.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_back.cpu().numpy().reshape(Q, K), kmers)
    kmers[7, 2] = ord("n")
    kmers[9, 20] = ord("$")
    d_chars = torch.from_numpy(kmers.reshape(-1)).cuda()
    assert g.pack_device(d_chars.data_ptr(), K, Q, d_packed.data_ptr()) == 2
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("ordered", [0, 1])
def test_device_resident_packed_search(oracle, awfm, require_gpu, wide, ordered):
    """awfmGpuSearchHitsPacked: seed-order path straight from the packed words (no ASCII anywhere), and the unpack +
    ASCII search it falls back to"""
    import torch
    n, K, Q = 250_000, 23, 12_345
    txt = synth.text(101, n)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 7)
    oi = oracle.Index.wrap(oracle.DNA, 8, 7, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(ordered)
    kmers = np.concatenate([synth.random_queries(102, Q // 2, K), synth.planted_queries(103, Q - Q // 2, K, txt)])
    chars, offsets = synth.fixed_csr(kmers)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    dev = torch.device("cuda")
    d_packed = torch.from_numpy(awfm.pack_kmers(kmers).view(np.int64)).to(dev)
    d_ranges = torch.full((Q * 2,), 7, dtype=torch.int64, device=dev)
    d_counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
    d_scratch = torch.zeros(Q * K + 8, dtype=torch.uint8, device=dev)
    g.search_hits_packed(d_packed.data_ptr(), K, Q, d_ranges.data_ptr(), d_counts.data_ptr(),
                         0 if ordered else d_scratch.data_ptr())
    torch.cuda.synchronize()
    ranges = d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2)
    hit = cnt > 0
    assert np.array_equal(d_counts.cpu().numpy().view(np.uint32), cnt)
    assert np.array_equal(ranges[hit, 0], sp[hit]) and np.array_equal(ranges[hit, 1], ep[hit])
    assert np.all(ranges[~hit, 0] > ranges[~hit, 1])
    if not ordered:  # searched as ASCII: without scratch the call must refuse, loudly
        with pytest.raises(awfm.AwFmError):
            g.search_hits_packed(d_packed.data_ptr(), K, Q, d_ranges.data_ptr(), d_counts.data_ptr(), 0)
    g.destroy()
    ix.dealloc()


def test_locate_into_page_locked_host_memory(oracle, awfm, require_gpu, monkeypatch):
    """awfmGpuLocateTo: the kernel that produces the positions stores them where the caller reads them (page-locked host
    memory), the walk's work array stays on the device"""
    import torch
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    n, K, Q = 150_000, 18, 9_000
    txt = synth.text(111, n)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 7)
    g = awfm.GpuIndex(ix)
    kmers = synth.planted_queries(112, Q, K, txt)
    cnt, pos = _oracle_answers(oracle, oracle.DNA, ix, 8, 7, kmers)
    dev = torch.device("cuda")
    d_chars = torch.from_numpy(kmers.reshape(-1).copy()).to(dev)
    d_ranges = torch.zeros(Q * 2, dtype=torch.int64, device=dev)
    g.search(d_chars.data_ptr(), 0, K, Q, d_ranges.data_ptr(), 0)
    d_hit_off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
    d_scratch = torch.zeros(awfm.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
    total = g.hit_offsets(d_ranges.data_ptr(), Q, d_hit_off.data_ptr(), d_scratch.data_ptr())
    assert total == len(pos)
    d_work = torch.zeros(total, dtype=torch.int64, device=dev)
    address = L.awfmGpuHostAlloc(total * 8)
    rc = L.awfmGpuLocateTo(g.handle, d_ranges.data_ptr(), d_hit_off.data_ptr(), Q, total, d_work.data_ptr(), address, None)
    assert rc == 1
    torch.cuda.synchronize()
    got = np.ctypeslib.as_array(C.cast(address, C.POINTER(C.c_uint64)), shape=(total,)).copy()
    assert np.array_equal(got, pos)
    L.awfmGpuHostFree(address)
    g.destroy()
    ix.dealloc()


def test_stream_edge_cases(oracle, awfm, require_gpu):
    """an empty batch calls no sink; chunks of one k-mer; a batch smaller than a chunk; amino k-mers beyond what a word
    holds and nucleotide k-mers beyond 32 characters are refused"""
    from avxwindowfmindex_amd import _lib
    n, K = 60_000, 12
    txt = synth.text(121, n)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 4, 5)
    g = awfm.GpuIndex(ix)
    calls = []
    g.stream(np.zeros(0, np.uint64), K, locate=True, sink=lambda *a: calls.append(a) or 0)
    assert calls == []
    kmers = synth.planted_queries(122, 7, K, txt)
    cnt, pos = _oracle_answers(oracle, oracle.DNA, ix, 4, 5, kmers)
    packed = awfm.pack_kmers(kmers)
    for chunk in (1, 3, 7, 1000):
        counts, positions = g.stream(packed, K, locate=True, chunk=chunk)
        assert np.array_equal(counts, cnt) and np.array_equal(positions, pos), chunk
    with pytest.raises(awfm.AwFmError):
        g.stream(packed, 33, locate=False)
    amino = awfm.create_index(synth.text(123, 20_000, synth.AMINO_ALPHABET), awfm.AwFmAlphabetAmino, 4, 2)
    ga = awfm.GpuIndex(amino)
    with pytest.raises(awfm.AwFmError):
        ga.stream(packed, 13, locate=False)
    for h in (g, ga):
        h.destroy()
    ix.dealloc()
    amino.dealloc()


def test_one_image_packed_then_ascii_and_growing_kmer_length(oracle, awfm, require_gpu):
    """the pipeline's slots are re-used across batches on one image: a slot sized for 8-byte packed words (or 10-byte
    ASCII k-mers) must grow before a batch of wider k-mers with the same chunk size is uploaded into it"""
    n, chunk = 200_000, 2048
    txt = synth.text(81, n)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 6)
    g = awfm.GpuIndex(ix)
    k21 = np.concatenate([synth.random_queries(82, 3000, 21), synth.planted_queries(83, 3000, 21, txt)])
    k10 = synth.planted_queries(84, 6000, 10, txt)
    k30 = np.concatenate([synth.random_queries(85, 3000, 30), synth.planted_queries(86, 3000, 30, txt)])
    want = {K: _oracle_answers(oracle, oracle.DNA, ix, 8, 6, q) for K, q in ((21, k21), (10, k10), (30, k30))}
    # (1) packed words first (8 B per k-mer in the slot), then ASCII 21-mers (21 B per k-mer), same chunk size
    counts, positions = g.stream(awfm.pack_kmers(k21), 21, locate=True, chunk=chunk)
    assert np.array_equal(counts, want[21][0]) and np.array_equal(positions, want[21][1])
    counts, positions = g.stream(k21.reshape(-1), 21, locate=True, chunk=chunk, packed=False)
    assert np.array_equal(counts, want[21][0]) and np.array_equal(positions, want[21][1])
    g.destroy()
    # (2) ASCII 10-mers, then ASCII 30-mers on a fresh image, same chunk size
    g = awfm.GpuIndex(ix)
    counts, positions = g.stream(k10.reshape(-1), 10, locate=True, chunk=chunk, packed=False)
    assert np.array_equal(counts, want[10][0]) and np.array_equal(positions, want[10][1])
    counts, positions = g.stream(k30.reshape(-1), 30, locate=True, chunk=chunk, packed=False)
    assert np.array_equal(counts, want[30][0]) and np.array_equal(positions, want[30][1])
    # and back down: the larger slot serves the narrower batch
    counts, positions = g.stream(awfm.pack_kmers(k21), 21, locate=True, chunk=chunk)
    assert np.array_equal(counts, want[21][0]) and np.array_equal(positions, want[21][1])
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("ordered,chunk", [(1, 0), (1, 2500), (0, 3000)])
def test_sparse_result_pipeline_matches_oracle(oracle, awfm, require_gpu, ordered, chunk):
    """awfmGpuStreamPackedSparse / CharsSparse: per chunk the k-mers with hits as a list {k-mer, hit offsets} + positions.
    Sparse chunks take the fused path (the seed-order search appends the list itself); a chunk in which more than 1/64 of
    the k-mers occur overflows that list and is redone from dense results, as are the chunks after it; chunks that do not
    take the seed-order path are always made from dense results.  Same lists either way."""
    n, K = 300_000, 20
    txt = synth.text(301, n)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    g = awfm.GpuIndex(ix)
    g.set_ordered(ordered)
    # first 20000 k-mers: 100 occur (sparse: 1/200); then 6000 of which half occur (dense: the list overflows)
    head = np.concatenate([synth.random_queries(302, 19_900, K), synth.planted_queries(303, 100, K, txt)])
    head = head[np.random.default_rng(9).permutation(len(head))]
    tail = np.concatenate([synth.random_queries(304, 3000, K), synth.planted_queries(305, 3000, K, txt)])
    tail = tail[np.random.default_rng(10).permutation(len(tail))]
    for kmers in (head, np.concatenate([head, tail])):
        cnt, pos = _oracle_answers(oracle, oracle.DNA, ix, 8, 8, kmers)
        want_ids = np.flatnonzero(cnt).astype(np.uint64)
        want_off = np.concatenate([[0], np.cumsum(cnt[cnt > 0])]).astype(np.uint64)
        for packed in (True, False):
            data = awfm.pack_kmers(kmers) if packed else kmers.reshape(-1)
            ids, off, positions = g.stream_sparse(data, K, locate=True, chunk=chunk, packed=packed)
            assert np.array_equal(ids, want_ids), "k-mers with hits"
            assert np.array_equal(off, want_off) and np.array_equal(positions, pos), "hit offsets / positions"
            ids, off, none = g.stream_sparse(data, K, locate=False, chunk=chunk, packed=packed)
            assert none is None and np.array_equal(ids, want_ids) and np.array_equal(off, want_off)
    # chunk order, and what a chunk's call carries
    calls = []

    def sink(user, first, m, num, hit_kmers, hit_offsets, p, total):
        calls.append((first, m, num, total, list(np.ctypeslib.as_array(hit_kmers, shape=(num,))) if num else []))
        return 0

    g.stream_sparse(awfm.pack_kmers(head), K, locate=True, chunk=4096, sink=sink)
    assert [c[0] for c in calls] == list(range(0, len(head), 4096)) and sum(c[1] for c in calls) == len(head)
    cnt, _ = _oracle_answers(oracle, oracle.DNA, ix, 8, 8, head)
    for first, m, num, total, rel in calls:
        assert num == int((cnt[first:first + m] > 0).sum()) and total == int(cnt[first:first + m].sum())
        assert rel == sorted(rel) and all(0 <= r < m for r in rel)
    g.destroy()
    ix.dealloc()
