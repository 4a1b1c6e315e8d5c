"""Hit-budgeted locate: a batch whose hit list exceeds what may be resident on the device ($AWFM_GPU_HIT_BUDGET_BYTES) is
located window by window -- windows may begin and end inside one k-mer's list -- and the result must be exactly the
reference's: every positionList complete and in BWT order (ref src/AwFmParallelSearch.c:315-387 grows each list on its
own, so no batch is too large for it)."""
import os

import numpy as np
import pytest

from avxwindowfmindex_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def tiny_budget():
    os.environ["AWFM_GPU_HIT_BUDGET_BYTES"] = "8192"  # 1024 hits resident: windows of 512
    yield 1024
    os.environ.pop("AWFM_GPU_HIT_BUDGET_BYTES")


def _case(oracle, awfm, seed):
    n = 150_000
    txt = synth.text(seed, n)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 4, 4)
    kmers = [bytes(q) for q in synth.planted_queries(seed + 1, 1500, 7, txt)]     # ~10 hits each
    kmers += [b"ac", b"t", b"acg"]                                                 # 9 k, 37 k and 2 k hits: far above a window
    kmers += [bytes(q) for q in synth.random_queries(seed + 2, 500, 12)]          # mostly none
    kmers += [b"gtt"]                                                              # the batch ends inside a long list
    oi = oracle.Index.wrap(oracle.DNA, 4, 4, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    chars, offsets = oracle.pack_queries(kmers)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    ho, pos, _ = oi.batch_locate(sp, ep)
    assert cnt.max() > 20_000 and int(ho[-1]) > 60_000
    return ix, kmers, chars, offsets, cnt, ho, pos


def test_flat_host_locate_in_windows(oracle, awfm, require_gpu, tiny_budget):
    ix, kmers, chars, offsets, cnt, ho, pos = _case(oracle, awfm, 201)
    g = awfm.GpuIndex(ix)
    ranges, hit_off, got = g.locate_host(chars, offsets)
    assert np.array_equal(hit_off, ho) and np.array_equal(got, pos)
    # the windows themselves: consecutive, none above half the budget, the queries named are the ones they cut
    windows = []

    def sink(user, qb, qe, hb, he, p):
        windows.append((qb, qe, hb, he, np.ctypeslib.as_array(p, shape=(he - hb,)).copy()))
        return 0

    hit_off2 = g.locate_host_windows(chars, offsets, sink)
    assert np.array_equal(hit_off2, ho)
    assert len(windows) > 100 and windows[0][2] == 0 and windows[-1][3] == int(ho[-1])
    for (qb, qe, hb, he, p), nxt in zip(windows, windows[1:] + [None]):
        assert 0 < he - hb <= tiny_budget // 2 and (nxt is None or nxt[2] == he)
        assert ho[qb] <= hb < ho[qb + 1] and ho[qe - 1] < he <= ho[qe]
        assert np.array_equal(p, pos[hb:he])
    g.destroy()
    ix.dealloc()


def test_aos_locate_in_windows(oracle, awfm, require_gpu, tiny_budget):
    ix, kmers, chars, offsets, cnt, ho, pos = _case(oracle, awfm, 211)
    lst = awfm.KmerSearchList(len(kmers))
    lst.fill(kmers)
    awfm.parallel_search_locate(ix, lst, 4)
    assert np.array_equal(lst.counts(), cnt)
    caps = lst.capacities()
    assert np.array_equal(caps, np.maximum(cnt, 4))  # ref setPositionListCount: grown to exactly count, never shrunk
    for i in range(len(kmers)):
        assert np.array_equal(lst.positions(i), pos[ho[i]:ho[i + 1]]), i
    lst.dealloc()
    ix.dealloc()


@pytest.mark.parametrize("chunk", [0, 700])
def test_stream_locate_in_windows(oracle, awfm, require_gpu, tiny_budget, chunk):
    n, K = 150_000, 6
    txt = synth.text(221, n)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 4, 4)
    g = awfm.GpuIndex(ix)
    kmers = np.concatenate([synth.planted_queries(222, 2000, K, txt), synth.random_queries(223, 500, K)])
    kmers[1234] = np.frombuffer(b"aaaaaa", np.uint8)
    oi = oracle.Index.wrap(oracle.DNA, 4, 4, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    c, o = synth.fixed_csr(kmers)
    sp, ep, cnt, _ = oi.batch_search(c, o)
    ho, pos, _ = oi.batch_locate(sp, ep)
    assert int(ho[-1]) > 50_000  # ~37 hits per 6-mer: every chunk is far above the budget
    calls = []

    def sink(user, first, m, counts, positions, total):
        calls.append((first, m, np.ctypeslib.as_array(counts, shape=(m,)).copy(),
                      np.ctypeslib.as_array(positions, shape=(total,)).copy() if total else np.zeros(0, np.uint64)))
        return 0

    g.stream(awfm.pack_kmers(kmers), K, locate=True, chunk=chunk, sink=sink)
    assert len(calls) > 50
    assert np.array_equal(np.concatenate([p for *_, p in calls]), pos), "the calls' positions concatenate to the flat list"
    at = 0
    for first, m, counts, p in calls:  # groups of whole k-mers in order; every group fits a window
        assert first == at and np.array_equal(counts, cnt[first:first + m]) and p.size == int(ho[first + m] - ho[first])
        assert p.size <= tiny_budget // 2
        at += m
    assert at == len(kmers)
    counts, positions = g.stream(awfm.pack_kmers(kmers), K, locate=True, chunk=chunk)  # the gathering wrapper
    assert np.array_equal(counts, cnt) and np.array_equal(positions, pos)
    g.destroy()
    ix.dealloc()


def test_stream_slices_a_kmer_above_the_budget(oracle, awfm, require_gpu, tiny_budget):
    n, K = 150_000, 2
    txt = synth.text(231, n)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 4, 2)
    g = awfm.GpuIndex(ix)
    kmers = synth.random_queries(232, 40, K)
    oi = oracle.Index.wrap(oracle.DNA, 4, 2, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    c, o = synth.fixed_csr(kmers)
    sp, ep, cnt, _ = oi.batch_search(c, o)
    ho, pos, _ = oi.batch_locate(sp, ep)
    assert cnt.min() > 5000
    calls = []

    def sink(user, first, m, counts, positions, total):
        calls.append((first, m, int(counts[0]), total))
        return 0

    g.stream(awfm.pack_kmers(kmers), K, locate=True, chunk=0, sink=sink)
    assert all(m == 1 and 0 < total <= tiny_budget // 2 for _, m, _, total in calls)
    for i in range(len(kmers)):  # the slices of k-mer i are consecutive, carry its full count, and add up to it
        mine = [c for c in calls if c[0] == i]
        assert mine and all(c[2] == cnt[i] for c in mine) and sum(c[3] for c in mine) == cnt[i]
    counts, positions = g.stream(awfm.pack_kmers(kmers), K, locate=True, chunk=0)
    assert np.array_equal(counts, cnt) and np.array_equal(positions, pos)
    g.destroy()
    ix.dealloc()
