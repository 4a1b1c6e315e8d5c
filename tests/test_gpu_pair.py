"""The pair image (csrc/awfm_pair.h: two backward / LF steps per block read) against the CPU oracle.

Every nucleotide image carries it by default, so the whole GPU suite already runs through it; here it is switched
off and on over the same image and stressed where it has special cases: blocks flagged because of ambiguity letters
or the sentinel (stepped letter by letter), odd and even numbers of extension steps, walks that must stop at the
position in between two steps, and several superblocks of 2^23 positions.  Results must be those of the
letter-by-letter steps of ref src/AwFmSearch.c:42-103, :369-427 -- i.e. the oracle's -- bit for bit."""
import numpy as np
import pytest

from avxwindowfmindex_amd import synth

pytestmark = pytest.mark.gpu


def _search_and_locate(g, awfm, chars, offsets, fixed_length, Q):
    import torch
    dev = torch.device("cuda")
    d_chars = torch.from_numpy(np.concatenate([chars, np.zeros(8, np.uint8)])).to(dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev) if fixed_length == 0 else None
    d_ranges = torch.full((Q * 2,), 7, dtype=torch.int64, device=dev)
    d_counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
    g.search_hits(d_chars.data_ptr(), d_off.data_ptr() if d_off is not None else 0, fixed_length, Q, d_ranges.data_ptr(),
                  d_counts.data_ptr())
    d_hit_off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
    d_scratch = torch.zeros(awfm.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
    total = g.hit_offsets(d_ranges.data_ptr(), Q, d_hit_off.data_ptr(), d_scratch.data_ptr())
    d_pos = torch.zeros(max(total, 1), dtype=torch.int64, device=dev)
    g.locate(d_ranges.data_ptr(), d_hit_off.data_ptr(), Q, total, d_pos.data_ptr())
    torch.cuda.synchronize()
    return (d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2), d_counts.cpu().numpy().view(np.uint32),
            d_hit_off.cpu().numpy().view(np.uint64), d_pos[:total].cpu().numpy().view(np.uint64))


def _check(result, sp, ep, cnt, hit_off, pos):
    ranges, counts, ho, p = result
    hit = cnt > 0
    assert np.array_equal(counts, cnt), "counts differ"
    assert np.array_equal(ranges[hit, 0], sp[hit]) and np.array_equal(ranges[hit, 1], ep[hit]), "ranges of hits differ"
    assert np.all(ranges[~hit, 0] > ranges[~hit, 1])
    assert np.array_equal(ho, hit_off), "hit offsets differ"
    assert np.array_equal(p, pos), "positions differ"


@pytest.mark.parametrize("ratio,seed_k,K", [(8, 8, 21), (8, 8, 20), (5, 6, 9), (16, 4, 5), (3, 10, 11), (255, 8, 32)])
def test_pair_steps_on_a_text_full_of_ambiguity_runs(oracle, awfm, require_gpu, wide, ratio, seed_k, K):
    """flagged blocks everywhere (an 'n' run every few thousand characters, sanitised to x), odd and even step counts,
    sampling ratios that are and are not powers of two; the same image with and without its pair blocks"""
    n = 400_000
    txt = synth.text(81 + K, n).copy()
    rng = np.random.default_rng(K)
    for start in rng.integers(0, n - 100, 150):
        txt[start:start + rng.integers(1, 60)] = ord("n")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, ratio, seed_k)
    oi = oracle.Index.wrap(oracle.DNA, ratio, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                           ix.packed_sa())
    g = awfm.GpuIndex(ix)
    assert g.has_pair_image
    g.set_ordered(1)
    Q = 20011
    q = np.concatenate([synth.random_queries(82, Q // 3, K), synth.planted_queries(83, Q - Q // 3, K, txt)]).copy()
    chars, offsets = synth.fixed_csr(q)  # planted k-mers carry the text's n runs: searched as x by the general kernel
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    assert cnt.sum() > Q // 2
    _check(_search_and_locate(g, awfm, chars, offsets, K, Q), sp, ep, cnt, hit_off, pos)
    g.set_pair_image(False)
    assert not g.has_pair_image
    _check(_search_and_locate(g, awfm, chars, offsets, K, Q), sp, ep, cnt, hit_off, pos)
    g.set_pair_image(True)
    assert g.has_pair_image
    _check(_search_and_locate(g, awfm, chars, offsets, K, Q), sp, ep, cnt, hit_off, pos)
    g.destroy()
    ix.dealloc()


def test_pair_steps_in_mixed_length_batches_and_repetitive_texts(oracle, awfm, require_gpu, wide):
    """CSR batches (every k-mer its own number of steps, the odd one first) over a repetitive text: long hit lists,
    long LF chains that end on either parity, the sentinel's block"""
    txt = np.frombuffer((b"acgtacgtaa" * 20000) + b"ttttttttttttttttttttt" + (b"gattaca" * 3000), np.uint8)
    for ratio in (1, 2, 7, 64):
        ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, ratio, 5)
        oi = oracle.Index.wrap(oracle.DNA, ratio, 5, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
        g = awfm.GpuIndex(ix)
        g.set_ordered(1)
        Q = 6000
        chars, offsets = synth.mixed_queries(90 + ratio, Q, txt, synth.DNA_ALPHABET, 4, 33)
        sp, ep, cnt, _ = oi.batch_search(chars, offsets)
        keep = np.cumsum(cnt) < 3_000_000  # bound the hit list
        Q = int(keep.sum())
        offsets = offsets[: Q + 1]
        chars = chars[: int(offsets[-1])]
        sp, ep, cnt = sp[:Q], ep[:Q], cnt[:Q]
        hit_off, pos, _ = oi.batch_locate(sp, ep)
        for enable in (True, False):
            g.set_pair_image(enable)
            _check(_search_and_locate(g, awfm, chars, offsets, 0, Q), sp, ep, cnt, hit_off, pos)
        g.destroy()
        ix.dealloc()


def test_pair_image_env_knob_and_device_bytes(oracle, awfm, require_gpu, monkeypatch):
    n = 300_000
    txt = synth.text(95, n)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    with_pair = awfm.GpuIndex(ix)
    monkeypatch.setenv("AWFM_GPU_PAIR", "0")
    without = awfm.GpuIndex(ix)
    assert with_pair.has_pair_image and not without.has_pair_image
    blocks = (ix.bwt_length + 127) // 128
    assert with_pair.device_bytes - without.device_bytes == blocks * 128 + 20 * 12 + 128  # one superblock of 2^23 positions: 20 bases
    amino = awfm.create_index(synth.text(96, 50_000, synth.AMINO_ALPHABET), awfm.AwFmAlphabetAmino, 8, 3)
    monkeypatch.delenv("AWFM_GPU_PAIR")
    ga = awfm.GpuIndex(amino)
    assert not ga.has_pair_image  # nucleotide only
    for h in (with_pair, without, ga):
        h.destroy()
    ix.dealloc()
    amino.dealloc()


def test_pair_image_over_several_superblocks(oracle, awfm, require_gpu):
    """2^23 positions per superblock of the pair image: a 40 Mbp index (built on the GPU) has five; located planted
    k-mers must come back at their planting offsets, counts must equal the oracle's on a sample"""
    import torch
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    n, K, Q = 40_000_000, 24, 2_000_000
    dev = torch.device("cuda")
    d_text = torch.empty(n, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 97, 0, None) == 1
    ix = awfm.gpu_create_index(d_text.data_ptr(), awfm.AwFmAlphabetDna, 8, 10, on_device_length=n)
    g = awfm.GpuIndex(ix, acquire=True)
    assert g.has_pair_image
    g.set_ordered(1)
    d_chars = torch.empty(Q * K, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthPlantedQueries(d_chars.data_ptr(), 0, Q, K, 98, d_text.data_ptr(), n, None) == 1
    d_ranges = torch.empty(Q * 2, dtype=torch.int64, device=dev)
    d_counts = torch.empty(Q, dtype=torch.int32, device=dev)
    g.search_hits(d_chars.data_ptr(), 0, K, Q, d_ranges.data_ptr(), d_counts.data_ptr())
    d_hit_off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
    d_scratch = torch.zeros(awfm.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
    total = g.hit_offsets(d_ranges.data_ptr(), Q, d_hit_off.data_ptr(), d_scratch.data_ptr())
    d_pos = torch.zeros(total, dtype=torch.int64, device=dev)
    g.locate(d_ranges.data_ptr(), d_hit_off.data_ptr(), Q, total, d_pos.data_ptr())
    torch.cuda.synchronize()
    counts = d_counts.cpu().numpy().view(np.uint32)
    assert counts.min() >= 1
    # every planted k-mer is found where it was planted
    planted_at = synth.planted_offsets(98, Q, K, n)
    hit_off = d_hit_off.cpu().numpy().view(np.uint64)
    pos = d_pos.cpu().numpy().view(np.uint64)
    single = counts == 1
    assert single.mean() > 0.99
    assert np.array_equal(pos[hit_off[:-1][single]], planted_at[single])
    # the same search without the pair image: identical ranges, offsets and positions
    g.set_pair_image(False)
    d_ranges2 = torch.empty(Q * 2, dtype=torch.int64, device=dev)
    g.search_hits(d_chars.data_ptr(), 0, K, Q, d_ranges2.data_ptr(), 0)
    d_pos2 = torch.zeros(total, dtype=torch.int64, device=dev)
    g.locate(d_ranges2.data_ptr(), d_hit_off.data_ptr(), Q, total, d_pos2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(d_ranges2, d_ranges) and torch.equal(d_pos2, d_pos)
    # and the oracle on a sample
    m = 50_000
    oi = oracle.Index.wrap(oracle.DNA, 8, 10, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    chars = d_chars[: m * K].cpu().numpy()
    sp, ep, cnt, _ = oi.batch_search(chars, np.arange(m + 1, dtype=np.uint64) * np.uint64(K))
    ho, opos, _ = oi.batch_locate(sp, ep)
    assert np.array_equal(counts[:m], cnt) and np.array_equal(pos[: int(ho[-1])], opos)
    ix.dealloc()


@pytest.mark.parametrize("ordered", [1, 0])
def test_search_hits_sparse_writes_counts_for_all_and_ranges_for_hits(oracle, awfm, require_gpu, wide, ordered):
    """awfmGpuSearchHitsSparse (include/awfm_gpu.h): a count for every k-mer, the exact range for every k-mer with hits,
    and the range of a k-mer without hits possibly untouched (the seed-order path leaves it as passed; the general
    kernel writes its exact empty range); hit offsets from the counts and positions on top are those of the oracle"""
    import torch
    n, K, Q = 600_000, 19, 30011
    txt = synth.text(131, n)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 9)
    oi = oracle.Index.wrap(oracle.DNA, 8, 9, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(ordered)
    q = np.concatenate([synth.random_queries(132, Q // 2, K), synth.planted_queries(133, Q - Q // 2, K, txt)]).copy()
    rng = np.random.default_rng(5)
    rng.shuffle(q, axis=0)
    chars, offsets = synth.fixed_csr(q)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    dev = torch.device("cuda")
    d_chars = torch.from_numpy(np.concatenate([chars, np.zeros(8, np.uint8)])).to(dev)
    d_ranges = torch.full((Q * 2,), 7, dtype=torch.int64, device=dev)
    d_counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
    g.search_hits_sparse(d_chars.data_ptr(), 0, K, Q, d_ranges.data_ptr(), d_counts.data_ptr())
    d_hit_off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
    d_scratch = torch.zeros(awfm.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
    total = g.hit_offsets_from_counts(d_counts.data_ptr(), Q, d_hit_off.data_ptr(), d_scratch.data_ptr())
    d_pos = torch.zeros(max(total, 1), dtype=torch.int64, device=dev)
    g.locate(d_ranges.data_ptr(), d_hit_off.data_ptr(), Q, total, d_pos.data_ptr())
    torch.cuda.synchronize()
    ranges = d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2)
    hit = cnt > 0
    assert 0.3 < hit.mean() < 0.7
    assert np.array_equal(d_counts.cpu().numpy().view(np.uint32), cnt)
    assert np.array_equal(ranges[hit, 0], sp[hit]) and np.array_equal(ranges[hit, 1], ep[hit])
    untouched = (ranges[~hit] == 7).all(axis=1)
    empty = ranges[~hit, 0] > ranges[~hit, 1]
    assert np.all(untouched | empty)
    if ordered:
        assert untouched.all(), "the seed-order path does not write the ranges of k-mers without hits"
    assert np.array_equal(d_hit_off.cpu().numpy().view(np.uint64), hit_off)
    assert np.array_equal(d_pos[:total].cpu().numpy().view(np.uint64), pos)
    # counts are what says which ranges were written: they are required
    with pytest.raises(awfm.AwFmError):
        g.search_hits_sparse(d_chars.data_ptr(), 0, K, Q, d_ranges.data_ptr(), 0)
    g.destroy()
    ix.dealloc()
