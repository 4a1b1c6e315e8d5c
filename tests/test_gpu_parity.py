"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Mirrors the properties of the reference's test/parallelSearch/parallelSearchTest.c:45-456,
test/inMemorySaTest/inMemorySaTest.c:29-266 and test/searchTest/searchTest.c:124-200, with
seeded inputs and exact {sp,ep}/hit-order comparison instead of order-insensitive checks.
"""
import os

import numpy as np
import pytest

from avxwindowfmindex_amd import synth

pytestmark = pytest.mark.gpu


def _mixed_queries(seed, count, txt, alphabet, lo, hi, ambiguity=None, upper=False):
    chars, offsets = synth.mixed_queries(seed, count, txt, alphabet, lo, hi)
    chars = chars.copy()
    rng = np.random.default_rng(seed)
    if ambiguity is not None and chars.size:
        hit = rng.random(chars.size) < 0.01
        chars[hit] = ambiguity
    if upper and chars.size:
        up = rng.random(chars.size) < 0.3
        chars[up] = chars[up] & 0xDF
    return chars, offsets


def _check_against_oracle(O, awfm, txt, alpha, oalpha, ratio, seed_k, chars, offsets):
    ix = awfm.create_index(txt, alpha, ratio, seed_k)
    oi = O.Index.wrap(oalpha, ratio, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                      ix.packed_sa())
    g = awfm.GpuIndex(ix)
    assert g.is_wide == (os.environ.get("AWFM_GPU_FORCE_WIDE", "0") == "1")  # the `wide` fixture really took effect
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    ranges, counts = g.count_host(chars, offsets)
    assert np.array_equal(ranges[:, 0], sp), "sp differs"
    assert np.array_equal(ranges[:, 1], ep), "ep differs"
    assert np.array_equal(counts, cnt), "count differs"
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    r2, ho2, pos2 = g.locate_host(chars, offsets)
    assert np.array_equal(r2, ranges)
    assert np.array_equal(ho2, hit_off), "hit offsets differ"
    assert np.array_equal(pos2, pos), "positions differ (BWT order)"
    g.destroy()
    ix.dealloc()
    return int(hit_off[-1])


@pytest.mark.parametrize("n,ratio,seed_k", [(29, 1, 3), (4096, 3, 4), (65536, 8, 8), (300000, 16, 6), (100000, 255, 5)])
def test_dna_parity(oracle, awfm, require_gpu, wide, n, ratio, seed_k):
    txt = synth.text(n + 7, n, synth.DNA_ALPHABET).copy()
    if n > 100:
        txt[10:14] = ord("n")  # ambiguity run in the text (sanitised to 'x')
        txt[n // 2] = ord("N")
    chars, offsets = _mixed_queries(1000 + n, 4000, txt, synth.DNA_ALPHABET, 1, min(40, n), ambiguity=ord("x"),
                                    upper=True)
    hits = _check_against_oracle(oracle, awfm, txt, awfm.AwFmAlphabetDna, oracle.DNA, ratio, seed_k, chars, offsets)
    assert hits > 0


@pytest.mark.parametrize("n,ratio,seed_k", [(31, 1, 1), (5000, 3, 2), (60000, 8, 3), (200000, 16, 4)])
def test_amino_parity(oracle, awfm, require_gpu, n, ratio, seed_k):
    txt = synth.text(n + 3, n, synth.AMINO_ALPHABET).copy()
    if n > 100:
        txt[20:23] = ord("x")  # sanitised to 'z'
        txt[n // 3] = ord("b")
    chars, offsets = _mixed_queries(2000 + n, 4000, txt, synth.AMINO_ALPHABET, 1, min(25, n), ambiguity=ord("z"))
    hits = _check_against_oracle(oracle, awfm, txt, awfm.AwFmAlphabetAmino, oracle.AMINO, ratio, seed_k, chars,
                                 offsets)
    assert hits > 0


def test_fixed_length_and_empty_batch(oracle, awfm, require_gpu):
    txt = synth.text(5, 50000)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                           ix.packed_sa())
    g = awfm.GpuIndex(ix)
    q = np.concatenate([synth.random_queries(9, 3000, 12), synth.planted_queries(10, 3000, 12, txt)])
    chars, offsets = synth.fixed_csr(q)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    ranges, counts = g.count_host(chars, None, fixed_length=12)
    assert np.array_equal(ranges[:, 0], sp) and np.array_equal(ranges[:, 1], ep) and np.array_equal(counts, cnt)
    # empty batch is a no-op
    r0, c0 = g.count_host(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert r0.shape == (0, 2) and c0.size == 0
    # zero-length k-mers (documented UB in the reference) come back as empty ranges
    r1, c1 = g.count_host(np.frombuffer(b"acgt", np.uint8), np.array([0, 0, 4, 4], np.uint64))
    assert c1[0] == 0 and c1[2] == 0 and c1[1] == oi.search_list([b"acgt"])[2][0]
    g.destroy()
    ix.dealloc()


def test_repetitive_text_many_hits(oracle, awfm, require_gpu, wide):
    """long position lists (wave-cooperative expansion) and long LF chains, sentinel wrap included"""
    txt = np.frombuffer((b"acgtacgtaa" * 3000) + b"ttttttttttttttttttttt", np.uint8)
    for ratio in (1, 7, 200):
        ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, ratio, 4)
        oi = oracle.Index.wrap(oracle.DNA, ratio, 4, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                               ix.packed_sa())
        g = awfm.GpuIndex(ix)
        kmers = [b"a", b"acgt", b"acgtacgtaaacgt", b"tttt", b"gta", b"cccc", txt[:30].tobytes(), b"t" * 21, b"x"]
        chars, offsets = oracle.pack_queries(kmers)
        sp, ep, cnt, _ = oi.batch_search(chars, offsets)
        hit_off, pos, _ = oi.batch_locate(sp, ep)
        ranges, ho, p = g.locate_host(chars, offsets)
        assert np.array_equal(ranges[:, 0], sp) and np.array_equal(ranges[:, 1], ep)
        assert np.array_equal(ho, hit_off) and np.array_equal(p, pos)
        assert int(hit_off[-1]) > 10000
        g.destroy()
        ix.dealloc()


def test_long_hit_lists_located_in_windows_through_the_full_suffix_array(oracle, awfm, require_gpu, wide):
    """windows of LONG hit lists (64 hits per k-mer and more) on an image with the full suffix array: expandLongKernel, parallel
    over the hits -- chunks of 16384 hits that begin and end inside a list, lists of a few hits and k-mers without hits
    between lists of 10^4, windows that begin and end inside a list or hold a single hit, the whole list as one window --
    against the oracle's positions; 32-bit entries, and the 40-bit ones of an image that runs 64-bit positions (`wide`)"""
    import torch
    rng = np.random.default_rng(11)
    unit = rng.integers(0, 4, 700)
    body = np.frombuffer(b"acgt", np.uint8)[np.concatenate([unit] * 120 + [rng.integers(0, 4, 30000)])]
    txt = np.concatenate([body, np.frombuffer(b"t" * 40, np.uint8)]).copy()
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 4)
    oi = oracle.Index.wrap(oracle.DNA, 8, 4, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_dense_sa(True)
    kmers = []
    for i in range(600):  # 1..5-mers (10^3..10^4 hits), windows of the repeat unit (120 hits), random 12-mers (none), a few long ones
        r = i % 6
        if r < 3:
            kmers.append(bytes(np.frombuffer(b"acgt", np.uint8)[rng.integers(0, 4, 1 + (i % 5))]))
        elif r == 3:
            at = int(rng.integers(0, 600))
            kmers.append(txt[at:at + 20].tobytes())
        elif r == 4:
            kmers.append(bytes(np.frombuffer(b"acgt", np.uint8)[rng.integers(0, 4, 12)]))
        else:
            at = int(rng.integers(84000, 110000))
            kmers.append(txt[at:at + 25].tobytes())
    chars, offsets = oracle.pack_queries(kmers)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    total, Q = int(hit_off[-1]), len(kmers)
    assert total > 64 * Q and (cnt == 0).sum() > 50 and ((cnt > 0) & (cnt < 4)).sum() > 50
    dev = torch.device("cuda")
    d_ranges = torch.from_numpy(np.stack([sp, ep], 1).astype(np.uint64).view(np.int64).reshape(-1)).to(dev)
    d_off = torch.from_numpy(hit_off.view(np.int64)).to(dev)
    cuts = sorted({0, total, 1, 16384, 16385, 40000, total - 1, total // 2, int(hit_off[7]), int(hit_off[7]) + 3, int(hit_off[300]) - 1})
    for b, e in list(zip(cuts[:-1], cuts[1:])) + [(0, total), (5, 6)]:
        qb = int(np.searchsorted(hit_off, b, side="right")) - 1
        qe = int(np.searchsorted(hit_off, e, side="left"))
        d_pos = torch.full((e - b + 8,), -1, dtype=torch.int64, device=dev)
        g.locate_window(d_ranges.data_ptr(), d_off.data_ptr(), max(qb, 0), min(max(qe, qb + 1), Q), b, e, d_pos.data_ptr())
        torch.cuda.synchronize()
        got = d_pos.cpu().numpy()
        assert np.array_equal(got[:e - b].view(np.uint64), pos[b:e]), (b, e)
        assert np.all(got[e - b:] == -1), "written behind the window"
    g.destroy()
    ix.dealloc()


def test_full_suffix_array_of_a_text_with_long_runs(oracle, awfm, require_gpu, monkeypatch, wide, diag):
    """A text with R long runs of one letter, R a multiple of the sampling ratio (a genome's runs of N): the suffixes inside
    the runs move R places per LF step and never meet a sample until a run ends.  The AUTOMATIC construction of the full
    suffix array parks such walks after 32 x ratio steps and completes the parked entries from each other (pointer
    jumping); a construction that was asked for walks to the end; an index built on the GPU hands its own suffix array to
    its image and walks nothing; without the array the locate walks, as the reference does.  Positions against the oracle
    in all four cases -- k-mers right behind a run (their walks enter it) and k-mers of N (hits INSIDE the runs) included.
    Round 6: with 64-bit positions (`wide`) the array has 40-bit entries and its own construction (64-bit entries, the parked
    walks in a list read from the round before)."""
    import torch
    n, runs, run_len, ratio = 160000, 8, 3000, 8
    txt = synth.text(n + 71, n, synth.DNA_ALPHABET).copy()
    starts = [5000 + i * 19000 for i in range(runs)]
    for at in starts:
        txt[at:at + run_len] = ord("n")
    kmers = [txt[at + run_len: at + run_len + 14].tobytes() for at in starts]  # right behind a run: the walk enters it
    kmers += [txt[at - 14: at].tobytes() for at in starts] + [bytes(r) for r in synth.planted_queries(72, 400, 16, txt)]
    kmers = [k for k in kmers if b"n" not in k] + [b"n" * 12, b"n" * 31]
    chars, offsets = oracle.pack_queries(kmers)

    def check(ix, expect_dense):
        oi = oracle.Index.wrap(oracle.DNA, ratio, 6, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
        sp, ep, cnt, _ = oi.batch_search(chars, offsets)
        hit_off, pos, _ = oi.batch_locate(sp, ep)
        g = awfm.GpuIndex(ix, acquire=True)
        assert g.has_dense_sa == expect_dense
        ranges, ho, p = g.locate_host(chars, offsets)
        assert np.array_equal(ho, hit_off) and np.array_equal(p, pos) and len(pos) >= len(kmers)
        return g

    monkeypatch.setenv("AWFM_GPU_DENSE_SA", "auto")
    for park_list in (None, "0", "1000"):  # the parked walks in a list (round 5) / an entry per position / a list that overflows
        if park_list is not None:
            diag(park_list=park_list)
        ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, ratio, 6)
        check(ix, True)  # 7/8 of the positions inside the runs parked, then completed
        ix.dealloc()
    diag(park_list=None)
    monkeypatch.setenv("AWFM_GPU_DENSE_SA", "0")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, ratio, 6)
    check(ix, False)  # the walk at query time
    ix.dealloc()
    monkeypatch.setenv("AWFM_GPU_DENSE_SA", "1")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, ratio, 6)
    check(ix, True)  # asked for: walked to the end
    ix.dealloc()
    monkeypatch.setenv("AWFM_GPU_DENSE_SA", "auto")
    d_text = torch.from_numpy(txt).to(torch.device("cuda"))
    ix = awfm.gpu_create_index(d_text.data_ptr(), awfm.AwFmAlphabetDna, ratio, 6, on_device_length=n)
    g = check(ix, True)  # the builder's own array
    assert g.dense_sa_build_s < 0.05
    ix.dealloc()


def test_drop_in_aos_api(oracle, awfm, require_gpu):
    """awFmCreateKmerSearchList / awFmParallelSearchCount / awFmParallelSearchLocate exactly as a
    reference user calls them (ref test/parallelSearch/parallelSearchTest.c:105-214)"""
    txt = synth.text(77, 8000).copy()
    txt[100] = ord("x")
    raw = txt.tobytes()
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 9)
    chars, offsets = synth.mixed_queries(78, 1500, txt, synth.DNA_ALPHABET, 7, 36)
    kmers = [chars[int(offsets[i]):int(offsets[i + 1])].tobytes() for i in range(1500)]
    for threads in (1, 4):
        lst = awfm.KmerSearchList(1600)
        lst.fill(kmers)
        awfm.parallel_search_count(ix, lst, threads)
        counts = lst.counts()
        rc = awfm.parallel_search_locate(ix, lst, threads)
        assert rc == awfm.AwFmSuccess
        assert np.array_equal(lst.counts(), counts)
        caps = lst.capacities()
        assert np.all(caps >= counts) and np.all(caps[counts <= 4] == 4)
        for i in range(0, 1500, 7):
            k = kmers[i]
            expect = sorted(p for p in range(len(raw) - len(k) + 1) if raw[p:p + len(k)] == k)
            assert sorted(lst.positions(i).tolist()) == expect
            assert counts[i] == len(expect)
        lst.dealloc()
    ix.dealloc()


def test_sa_staged_from_file(oracle, awfm, require_gpu, tmp_path):
    """keepSuffixArrayInMemory=false: the device image stages the sampled SA from the .awfmi file"""
    txt = synth.text(91, 30000)
    path = str(tmp_path / "ondisk.awfmi")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 5, 6, keep_sa_in_memory=False, file_src=path)
    assert ix.packed_sa() is None
    oi = oracle.Index.from_text(txt.tobytes(), oracle.DNA, 5, 6)
    q = synth.planted_queries(92, 2000, 15, txt)
    chars, offsets = synth.fixed_csr(q)
    sp, ep, _, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    g = awfm.GpuIndex(ix)
    _, ho, p = g.locate_host(chars, offsets)
    assert np.array_equal(ho, hit_off) and np.array_equal(p, pos)
    g.destroy()
    ix.dealloc()
    # and through a re-read index
    ix2 = awfm.read_index_from_file(path, keep_sa_in_memory=False)
    g2 = awfm.GpuIndex(ix2)
    _, ho2, p2 = g2.locate_host(chars, offsets)
    assert np.array_equal(ho2, hit_off) and np.array_equal(p2, pos)
    g2.destroy()
    ix2.dealloc()


@pytest.mark.parametrize("seed_k,deep_k", [(3, 5), (6, 9), (8, 12), (12, 16)])  # 16: the depth GRCh38-sized images get (34 GB)
def test_device_deep_seed_table_keeps_results_bit_identical(oracle, awfm, require_gpu, wide, seed_k, deep_k):
    """the optional device-only deeper seed table must not change a single range or position, including
    queries shorter than it, ambiguity letters inside/outside the deep suffix and absent k-mers"""
    n = 250000
    txt = synth.text(300 + deep_k, n).copy()
    txt[1000:1004] = ord("n")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, seed_k)
    oi = oracle.Index.wrap(oracle.DNA, 8, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                           ix.packed_sa())
    chars, offsets = _mixed_queries(400 + deep_k, 6000, txt, synth.DNA_ALPHABET, 1, 40, ambiguity=ord("x"), upper=True)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    g = awfm.GpuIndex(ix)
    before = g.device_bytes
    g.set_deep_seed(deep_k)
    # {sp, length}: 8 bytes per entry below 2^32 positions; with the next-step bits (images with pair blocks) the lengths of
    # 65535 and more have a table of their own, a word per 2^15 positions
    # (round 6: an image that runs 64-bit positions has sp36 | length12 | next16 entries and 64-bit lengths of 4095 and more, a
    # word per 2^11 positions)
    side = 8 * ((ix.bwt_length >> 11) + 2) if g.is_wide else 4 * ((ix.bwt_length >> 15) + 5)
    assert g.device_bytes - before - 8 * 4 ** deep_k in ((side,) if g.is_wide else (0, side))
    ranges, ho, p = g.locate_host(chars, offsets)
    assert np.array_equal(ranges[:, 0], sp) and np.array_equal(ranges[:, 1], ep)
    assert np.array_equal(ho, hit_off) and np.array_equal(p, pos)
    g.set_deep_seed(0)
    ranges2, counts2 = g.count_host(chars, offsets)
    assert np.array_equal(ranges2, ranges) and np.array_equal(counts2, cnt) and g.device_bytes == before
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("seed_k,deep_k,lanes", [(2, 4, None), (3, 5, None), (1, 3, "g4"), (2, 5, "g4")])
def test_amino_device_deep_seed_table_keeps_results_bit_identical(oracle, awfm, require_gpu, wide, monkeypatch, diag, seed_k, deep_k, lanes):
    """the same for the amino alphabet (20^deep_k entries, the index of ref src/AwFmKmerTable.c:37-51 over the last deep_k
    characters): exact ranges -- the first empty range of an absent k-mer included --, counts and positions, for k-mers
    shorter than the table, ambiguity letters (z, x, b) inside and outside its characters, upper case; two and four
    lanes per k-mer"""
    if lanes:
        diag(kernel=lanes)
    n = 120000
    txt = synth.text(700 + deep_k, n, synth.AMINO_ALPHABET).copy()
    txt[500:503] = ord("x")
    txt[60000:100000] = np.frombuffer(b"acdefghikl" * 4000, np.uint8)  # a repeat: ranges of 4000 and more (12-bit lengths saturate)
    txt[20000:32000] = np.frombuffer(b"mnp" * 4000, np.uint8)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetAmino, 8, seed_k)
    oi = oracle.Index.wrap(oracle.AMINO, 8, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                           ix.packed_sa())
    chars, offsets = _mixed_queries(800 + deep_k, 6000, txt, synth.AMINO_ALPHABET, 1, 14, ambiguity=ord("z"), upper=True)
    odd = np.random.default_rng(deep_k).random(chars.size) < 0.01
    chars[odd] = np.frombuffer(b"xbXB", dtype=np.uint8)[np.random.default_rng(deep_k + 1).integers(0, 4, int(odd.sum()))]  # the other ambiguity letters
    # (characters that are no amino letters at all -- j, o, u, $ -- are outside what the reference defines: its table index
    # runs past the table, ref src/AwFmKmerTable.c:37-51, and so does the oracle's; the kernels keep such k-mers away from
    # both tables, but there is nothing to compare them with)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    assert (cnt == 0).sum() > 500 and (cnt > 0).sum() > 500
    g = awfm.GpuIndex(ix)
    before = g.device_bytes
    g.set_deep_seed(deep_k)
    # (8-byte entries {sp, length12 | next20 << 12}; the lengths of 4095 and more have a table of their own, a word per 2^11 positions;
    # round 6, an image that runs 64-bit positions: sp36 | length8 | next20, the lengths of 255 and more in 64-bit words per 2^7 positions)
    side = 8 * ((ix.bwt_length >> 7) + 2) if g.is_wide else 4 * ((ix.bwt_length >> 11) + 5)
    assert g.deep_seed_k == deep_k and g.device_bytes == before + 8 * 20 ** deep_k + side
    ranges, ho, p = g.locate_host(chars, offsets)
    assert np.array_equal(ranges[:, 0], sp) and np.array_equal(ranges[:, 1], ep)
    assert np.array_equal(ho, hit_off) and np.array_equal(p, pos)
    # fixed-length batches through the hits-only entry point (what the bench and the pipelines call)
    K = deep_k + 2
    q = np.concatenate([synth.random_queries(810 + deep_k, 3000, K, synth.AMINO_ALPHABET), synth.planted_queries(811, 3000, K, txt)])
    fchars, foffsets = synth.fixed_csr(q)
    fsp, fep, fcnt, _ = oi.batch_search(fchars, foffsets)
    franges, fcounts = g.count_host(fchars, foffsets)
    assert np.array_equal(franges[:, 0], fsp) and np.array_equal(franges[:, 1], fep) and np.array_equal(fcounts, fcnt)
    g.set_deep_seed(0)
    ranges2, counts2 = g.count_host(chars, offsets)
    assert np.array_equal(ranges2, ranges) and np.array_equal(counts2, cnt) and g.device_bytes == before
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("seed_k,deep_k,K,planted_share", [(2, 4, 6, 0.1), (3, 5, 10, 0.5), (2, 5, 5, 0.05), (1, 3, 15, 1.0), (2, 4, 16, 0.3)])
def test_amino_lookup_first_keeps_hits_bit_identical(oracle, awfm, require_gpu, wide, monkeypatch, seed_k, deep_k, K, planted_share):
    """aminoLookupSearchKernel forced on ($AWFM_GPU_AMINO_LOOKUP=1) for fixed-length amino batches: the deeper table is looked
    up while the k-mers are decoded, survivors are stepped out of LDS, what the kernel does not cover (characters that are not
    one of the 20 letters among the table's, more than 64 survivors in a round of 256: the all-planted case) goes to the
    general kernel through the leftover list.  Dense results, counts only and the list of hits against the oracle; every
    byte alignment of the batch; upper case, ambiguity letters and non-letters anywhere in the k-mers."""
    import torch
    monkeypatch.setenv("AWFM_GPU_AMINO_LOOKUP", "1")
    n = 150000
    txt = synth.text(900 + K, n, synth.AMINO_ALPHABET).copy()
    txt[700:703] = ord("x")
    txt[90000:96000] = np.frombuffer(b"wyvt" * 1500, np.uint8)  # a repeat: ranges of 1500 (the 8-bit lengths of the wide entries saturate)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetAmino, 8, seed_k)
    oi = oracle.Index.wrap(oracle.AMINO, 8, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_deep_seed(deep_k)
    Q = 40009
    m = int(Q * planted_share)
    q = np.concatenate([synth.random_queries(910 + K, Q - m, K, synth.AMINO_ALPHABET), synth.planted_queries(911 + K, m, K, txt)]).copy()
    rng = np.random.default_rng(K)
    q = q[rng.permutation(Q)]
    flat = q.reshape(-1)
    odd = rng.random(flat.size) < 0.002
    flat[odd] = np.frombuffer(b"zxbZXB", dtype=np.uint8)[rng.integers(0, 6, int(odd.sum()))]
    up = rng.random(flat.size) < 0.3
    flat[up] = flat[up] & 0xDF
    chars, offsets = synth.fixed_csr(q)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=4)
    assert (cnt > 0).sum() >= m // 2
    dev = torch.device("cuda")
    for mis in (0, 1, 2, 3):
        buf = torch.zeros(chars.size + 64, dtype=torch.uint8, device=dev)
        buf[mis:mis + chars.size] = torch.from_numpy(chars).to(dev)
        d_ranges = torch.full((Q * 2,), 7, dtype=torch.int64, device=dev)
        d_counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
        g.search_hits(buf.data_ptr() + mis, 0, K, Q, d_ranges.data_ptr(), d_counts.data_ptr())
        torch.cuda.synchronize()
        assert g.last_ordered_kernel_is_lookup()
        _check_hits_contract(d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2), d_counts.cpu().numpy().view(np.uint32), sp, ep, cnt)
    d_counts2 = torch.full((Q,), 7, dtype=torch.int32, device=dev)
    g.search_hits(buf.data_ptr() + 3, 0, K, Q, 0, d_counts2.data_ptr())  # counts only
    torch.cuda.synchronize()
    assert np.array_equal(d_counts2.cpu().numpy().view(np.uint32), cnt)
    # the batch at the very end of its buffer (the 16-byte loads of the last k-mers must not leave it)
    tail = torch.zeros(4096 + chars.size, dtype=torch.uint8, device=dev)
    tail[4096:] = torch.from_numpy(chars).to(dev)
    g.search_hits(tail.data_ptr() + 4096, 0, K, Q, 0, d_counts2.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_counts2.cpu().numpy().view(np.uint32), cnt)
    # the list of the k-mers with hits, sorted and located without a host wait
    cap = Q
    d_kmers = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_list = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
    d_num = torch.zeros(1, dtype=torch.int32, device=dev)
    g.search_hits_compact(buf.data_ptr() + 3, 0, K, Q, d_kmers.data_ptr(), d_list.data_ptr(), cap, d_num.data_ptr())
    g.sort_hits_on_device(d_kmers.data_ptr(), d_list.data_ptr(), cap, d_num.data_ptr(), Q)
    torch.cuda.synchronize()
    has = np.flatnonzero(cnt > 0)
    listed = int(d_num.item())
    assert listed == len(has)
    r = d_list[:listed * 2].cpu().numpy().view(np.uint64).reshape(listed, 2)
    assert np.array_equal(d_kmers[:listed].cpu().numpy().view(np.uint32), has)
    assert np.array_equal(r[:, 0], sp[has]) and np.array_equal(r[:, 1], ep[has])
    g.destroy()
    ix.dealloc()


def test_amino_lookup_first_is_chosen_by_a_sample_of_the_batch(oracle, awfm, require_gpu, wide, monkeypatch):
    """without $AWFM_GPU_AMINO_LOOKUP a batch of 2^20 amino k-mers or more is sampled: random 8-mers nearly all end at the
    deeper table -> aminoLookupSearchKernel; k-mers drawn from the text all survive it -> the general kernel; counts against
    the oracle either way"""
    import torch
    monkeypatch.delenv("AWFM_GPU_AMINO_LOOKUP", raising=False)
    monkeypatch.setenv("AWFM_GPU_LOOKUP_PREDICT", "0")  # every search by its own sample
    n, K, Q = 200000, 8, (1 << 20) + 7
    txt = synth.text(950, n, synth.AMINO_ALPHABET).copy()
    ix = awfm.create_index(txt, awfm.AwFmAlphabetAmino, 8, 3)
    oi = oracle.Index.wrap(oracle.AMINO, 8, 3, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_deep_seed(5)
    dev = torch.device("cuda")
    for name, q in (("random", synth.random_queries(951, Q, K, synth.AMINO_ALPHABET).copy()), ("planted", synth.planted_queries(952, Q, K, txt).copy())):
        q[::1000, 5] = ord("x")
        if name == "random":
            q[5::64] = synth.planted_queries(953, len(q[5::64]), K, txt)
        chars, offsets = synth.fixed_csr(q)
        _, _, cnt, _ = oi.batch_search(chars, offsets, threads=4)
        d_chars = torch.from_numpy(chars).to(dev)
        d_counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
        g.search_hits(d_chars.data_ptr(), 0, K, Q, 0, d_counts.data_ptr())
        torch.cuda.synchronize()
        assert g.last_ordered_kernel_is_lookup() == (name == "random"), name
        assert np.array_equal(d_counts.cpu().numpy().view(np.uint32), cnt), name
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("pair", ["1", "0"])
def test_deep_seed_table_next_step_bits_and_long_ranges(oracle, awfm, require_gpu, wide, monkeypatch, pair):
    """On images with pair blocks the 8-byte entries of the deeper table are {sp, length16 | next16 << 16}: the lengths
    that do not fit (a tandem repeat: six 5-mers with 120 000 occurrences each) come from deepBigBySp[sp >> 15], and the seed-order
    search drops k-mers by the next-step bits.  General kernels (exact ranges), the seed-order search (hits) and the
    positions against the oracle; without pair blocks ($AWFM_GPU_PAIR=0) the entries stay {sp, length}."""
    import torch
    monkeypatch.setenv("AWFM_GPU_PAIR", pair)
    unit = np.frombuffer(b"acgtta", dtype=np.uint8)
    txt = np.concatenate([np.tile(unit, 120000), synth.text(501, 200000)]).copy()
    txt[720100:720104] = ord("n")
    seed_k, deep_k = 3, 5
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, seed_k)
    oi = oracle.Index.wrap(oracle.DNA, 8, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                           ix.packed_sa())
    g = awfm.GpuIndex(ix)
    before = g.device_bytes
    g.set_deep_seed(deep_k)
    # (with the next-step bits, i.e. pair blocks: the table of the lengths of 65535 and more -- the six 5-mers of the repeat --,
    # a word per 2^15 positions)
    # (an image that runs 64-bit positions: sp36 | length12 | next16 entries, the lengths of 4095 and more in 64-bit words per
    # 2^11 positions, with or without pair blocks)
    side = 8 * ((ix.bwt_length >> 11) + 2) if g.is_wide else (4 * ((ix.bwt_length >> 15) + 5) if pair == "1" else 0)
    assert g.device_bytes == before + 8 * 4 ** deep_k + side
    # mixed lengths through the general kernels: exact ranges, k-mers shorter than the table included
    chars, offsets = _mixed_queries(502, 4000, txt, synth.DNA_ALPHABET, 1, 30, ambiguity=ord("x"), upper=True)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    ranges, counts = g.count_host(chars, offsets)
    assert np.array_equal(ranges[:, 0], sp) and np.array_equal(ranges[:, 1], ep) and np.array_equal(counts, cnt)
    # fixed lengths through the seed-order search: 7 = table + one pair step, 8 = + a single step, 12, 5 = the table alone
    g.set_ordered(1)
    dev = torch.device("cuda")
    for K in (7, 8, 12, 5, 6):
        Q = 20011
        q = np.concatenate([synth.random_queries(503 + K, Q // 2, K), synth.planted_queries(504 + K, Q - Q // 2, K, txt)]).copy()
        chars, offsets = synth.fixed_csr(q)
        sp, ep, cnt, _ = oi.batch_search(chars, offsets)
        assert (cnt > 65535).any() or K > 8
        d_chars = torch.from_numpy(chars).to(dev)
        d_ranges = torch.full((Q * 2,), 7, dtype=torch.int64, device=dev)
        d_counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
        g.search_hits(d_chars.data_ptr(), 0, K, Q, d_ranges.data_ptr(), d_counts.data_ptr())
        torch.cuda.synchronize()
        _check_hits_contract(d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2), d_counts.cpu().numpy().view(np.uint32), sp, ep, cnt)
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("alphabet_name,ratio", [("dna", 8), ("dna", 255), ("amino", 5)])
def test_device_dense_sa_keeps_positions_bit_identical(oracle, awfm, require_gpu, wide, alphabet_name, ratio):
    amino = alphabet_name == "amino"
    letters = synth.AMINO_ALPHABET if amino else synth.DNA_ALPHABET
    alpha, oalpha = (awfm.AwFmAlphabetAmino, oracle.AMINO) if amino else (awfm.AwFmAlphabetDna, oracle.DNA)
    txt = synth.text(500 + ratio, 120000, letters)
    ix = awfm.create_index(txt, alpha, ratio, 3)
    oi = oracle.Index.from_text(txt.tobytes(), oalpha, ratio, 3)
    chars, offsets = _mixed_queries(501, 4000, txt, letters, 2, 20)
    sp, ep, _, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    g = awfm.GpuIndex(ix)
    before = g.device_bytes
    g.set_dense_sa(True)
    # 32-bit entries; 40-bit ones, packed five bytes apiece, for an image that runs 64-bit positions (round 6)
    assert g.device_bytes == before + ((ix.bwt_length + 3) // 4 * 20 + 16 if g.is_wide else 4 * ix.bwt_length)
    _, ho, p = g.locate_host(chars, offsets)
    assert np.array_equal(ho, hit_off) and np.array_equal(p, pos)
    g.set_dense_sa(False)
    _, ho2, p2 = g.locate_host(chars, offsets)
    assert np.array_equal(ho2, hit_off) and np.array_equal(p2, pos) and g.device_bytes == before
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("alphabet_name,ratio,pair", [("dna", 16, "1"), ("dna", 13, "0"), ("amino", 16, "1")])
def test_a_walk_the_walk_kernel_gives_up_is_walked_on_exactly(oracle, awfm, require_gpu, wide, monkeypatch, diag, alphabet_name, ratio, pair):
    """The hand-over between walkKernel and finishKernel holds 23 bits of LF steps.  A walk that has not met a sample by then --
    a hit right behind a long run of one letter: a valid index -- used to be finished as if it stood on one (advisor, round
    4); now it is parked and finishKernel walks it on, one thread, to its sample.  $AWFM_GPU_DIAG walk_give_up=3 moves the limit
    to three steps, so that most walks of an ordinary text (ratio 13 / 16) take that path: positions must be the oracle's,
    through the pair image and without, amino, 32- and 64-bit positions."""
    diag(walk_give_up=3)
    monkeypatch.setenv("AWFM_GPU_PAIR", pair)
    amino = alphabet_name == "amino"
    letters = synth.AMINO_ALPHABET if amino else synth.DNA_ALPHABET
    alpha, oalpha = (awfm.AwFmAlphabetAmino, oracle.AMINO) if amino else (awfm.AwFmAlphabetDna, oracle.DNA)
    txt = synth.text(700 + ratio, 90000, letters).copy()
    if not amino:
        txt[40000:40300] = ord("n")  # a run of the ambiguity letter: flagged pair blocks, the X count
    ix = awfm.create_index(txt, alpha, ratio, 3)
    oi = oracle.Index.from_text(txt.tobytes(), oalpha, ratio, 3)
    chars, offsets = _mixed_queries(701, 6000, txt, letters, 2, 20)
    sp, ep, _, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    g = awfm.GpuIndex(ix)
    g.set_dense_sa(False)
    _, ho, p = g.locate_host(chars, offsets)
    assert np.array_equal(ho, hit_off) and np.array_equal(p, pos)
    # the full suffix array, asked for: capped walks + pointer jumping (the cap is 32 x ratio: nothing parks on this text),
    # every entry the sorted suffixes' position
    g.set_dense_sa(True)
    _, ho, p = g.locate_host(chars, offsets)
    assert np.array_equal(ho, hit_off) and np.array_equal(p, pos)
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("pair", ["1", "0"])
def test_exact_ranges_through_the_tables(oracle, awfm, require_gpu, wide, monkeypatch, pair):
    """awfmGpuSearch through exactLookupSearchKernel (round 5; forced here, large batches take it by themselves): every k-mer's
    final range -- for a k-mer without hits the reference's first empty range, not just some empty one -- from ONE table entry
    (the deeper table's, or the table of the k-mer's own length) and exact pair steps behind it.  Fixed lengths at, around and far
    beyond the deeper table's depth and below it; CSR batches of 0..40 characters with ambiguity characters; a tandem repeat
    whose 11-mers occur more than 65535 times (saturated 16-bit lengths: the side list); the same batches through the general
    kernel for comparison.  Against the oracle, for every k-mer."""
    import torch
    monkeypatch.setenv("AWFM_GPU_PAIR", pair)
    n = 300000
    txt = synth.text(n + 77, n, synth.DNA_ALPHABET).copy()
    txt[100000:240000] = np.frombuffer(b"ac" * 70000, np.uint8)
    txt[50000:50400] = ord("n")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_deep_seed(11)
    dev = torch.device("cuda")
    Q = 40000

    def check(chars, offsets, fixed, what):
        sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=4)
        m = len(offsets) - 1
        d_chars = torch.from_numpy(np.concatenate([chars, np.zeros(8, np.uint8)])).to(dev)
        d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
        for knob in ("1", "0"):
            monkeypatch.setenv("AWFM_GPU_EXACT_LOOKUP", knob)
            d_ranges = torch.full((2 * m,), 5, dtype=torch.int64, device=dev)
            d_counts = torch.full((m,), 5, dtype=torch.int32, device=dev)
            g.search(d_chars.data_ptr(), 0 if fixed else d_off.data_ptr(), fixed, m, d_ranges.data_ptr(), d_counts.data_ptr())
            torch.cuda.synchronize()
            r = d_ranges.cpu().numpy().view(np.uint64).reshape(m, 2)
            assert np.array_equal(r[:, 0], sp) and np.array_equal(r[:, 1], ep), (what, knob, int(np.flatnonzero((r[:, 0] != sp) | (r[:, 1] != ep))[0]))
            assert np.array_equal(d_counts.cpu().numpy().view(np.uint32), cnt), (what, knob)

    for K in (11, 12, 13, 14, 21, 32, 9, 3):
        q = np.concatenate([synth.random_queries(300 + K, Q // 2, K), synth.planted_queries(400 + K, Q // 2, K, txt)])
        q = q[np.random.default_rng(K).permutation(len(q))].copy()
        q[::97, K // 2] = ord("n")
        q[5::1000] = np.frombuffer((b"ac" * 16)[:K], np.uint8)  # the tandem repeat's own k-mers: ranges of 7 * 10^4
        chars, offsets = synth.fixed_csr(q)
        check(chars, offsets, K, f"fixed {K}")
    chars, offsets = _mixed_queries(901, Q, txt, synth.DNA_ALPHABET, 0, 40, ambiguity=ord("n"), upper=True)
    check(chars, offsets, 0, "csr 0..40")
    assert g.length_tables[0] == 8 * (4 ** 11 - 4) // 3  # (the forced mode built the tables of the lengths 1..10)
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("lanes,chunk,budget", [("0,0,0", 4096, None), ("0,0", 5000, 1 << 15), (None, 70001, None), ("0", 1500, None)])
def test_drop_in_api_takes_a_list_in_chunks(oracle, awfm, require_gpu, monkeypatch, lanes, chunk, budget):
    """awFmParallelSearchCount / Locate cut a list into chunks ($AWFM_GPU_AOS_CHUNK) that the lanes take in turn, packing
    and scattering with all the caller's threads while the other lanes are in their device stage.  The list holds
    stretches of one length (packed in one pass) and stretches of mixed lengths (CSR), so neighbouring chunks differ in
    kind; with a small hit budget the positions of a chunk arrive in several windows.  Position lists keep the capacity
    rule of the reference (ref src/AwFmParallelSearch.c:367-387): grown to exactly `count` only when too small."""
    txt = synth.text(188, 150000)
    n_fixed, n_mixed = 9000, 7013
    fixed = synth.planted_queries(189, n_fixed, 9, txt)
    chars_m, offsets_m = synth.mixed_queries(190, n_mixed, txt, synth.DNA_ALPHABET, 1, 30)
    kmers = [fixed[i].tobytes() for i in range(n_fixed // 2)]
    kmers += [chars_m[int(offsets_m[i]):int(offsets_m[i + 1])].tobytes() for i in range(n_mixed)]
    kmers += [fixed[i].tobytes() for i in range(n_fixed // 2, n_fixed)]
    n = len(kmers)
    chars = np.frombuffer(b"".join(kmers), dtype=np.uint8)
    offsets = np.concatenate([[0], np.cumsum([len(k) for k in kmers])]).astype(np.uint64)
    oi = oracle.Index.from_text(txt.tobytes(), oracle.DNA, 8, 7)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    if lanes is None:
        monkeypatch.delenv("AWFM_GPU_DEVICES", raising=False)
    else:
        monkeypatch.setenv("AWFM_GPU_DEVICES", lanes)
    monkeypatch.setenv("AWFM_GPU_AOS_CHUNK", str(chunk))
    if budget:
        monkeypatch.setenv("AWFM_GPU_HIT_BUDGET_BYTES", str(budget))
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 7)
    lst = awfm.KmerSearchList(n)
    lst.fill(kmers)
    for threads in (7, 1):
        awfm.parallel_search_count(ix, lst, threads)
        assert np.array_equal(lst.counts(), cnt)
        assert awfm.parallel_search_locate(ix, lst, threads) == awfm.AwFmSuccess
        assert np.array_equal(lst.counts(), cnt)
        assert np.array_equal(lst.capacities(), np.maximum(cnt, 4))
        for i in list(range(0, n, 7)) + [n - 1]:
            assert np.array_equal(lst.positions(i), pos[int(hit_off[i]):int(hit_off[i + 1])]), i
    lst.dealloc()
    ix.dealloc()


def test_drop_in_api_shards_over_device_images(oracle, awfm, require_gpu, monkeypatch):
    """AWFM_GPU_DEVICES lists the devices awFmParallelSearch* shard a batch over (one host thread per entry,
    contiguous shards, no exchange; one index replica per distinct device, a device named again gets a lane
    with its own staging on the replica it already has).  With one GPU on the box the list names it three
    times: one image + two lanes, three shards, results identical to the unsharded run, also after the
    device-only accelerators are switched on and off on the primary image."""
    txt = synth.text(88, 120000)
    chars, offsets = synth.mixed_queries(89, 5003, txt, synth.DNA_ALPHABET, 3, 30)
    kmers = [chars[int(offsets[i]):int(offsets[i + 1])].tobytes() for i in range(5003)]
    oi = oracle.Index.from_text(txt.tobytes(), oracle.DNA, 8, 7)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    monkeypatch.setenv("AWFM_GPU_DEVICES", "0,0,0")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 7)
    lst = awfm.KmerSearchList(5003)
    lst.fill(kmers)
    awfm.parallel_search_count(ix, lst, 6)
    assert np.array_equal(lst.counts(), cnt)
    assert awfm.parallel_search_locate(ix, lst, 6) == awfm.AwFmSuccess
    assert np.array_equal(lst.counts(), cnt)
    for i in range(0, 5003, 11):
        assert np.array_equal(lst.positions(i), pos[int(hit_off[i]):int(hit_off[i + 1])])
    from avxwindowfmindex_amd import _lib
    import ctypes as C
    imgs = (C.c_void_p * 8)()
    L = _lib.lib()
    assert L.awfmGpuIndexAcquireAll(ix.ptr, imgs, 8) == 3 and len({imgs[0], imgs[1], imgs[2]}) == 3
    L.awfmGpuIndexDeviceBytes.restype = C.c_uint64
    assert L.awfmGpuIndexDeviceBytes(C.c_void_p(imgs[0])) > 0
    assert L.awfmGpuIndexDeviceBytes(C.c_void_p(imgs[1])) == 0 and L.awfmGpuIndexDeviceBytes(C.c_void_p(imgs[2])) == 0
    # accelerators are set on the primary and reach the lanes; a lane refuses them
    assert L.awfmGpuIndexSetDeepSeed(C.c_void_p(imgs[1]), 9) < 0
    for deep_k, dense in ((9, 1), (0, 0)):
        assert L.awfmGpuIndexSetDeepSeed(C.c_void_p(imgs[0]), deep_k) == awfm.AwFmSuccess
        assert L.awfmGpuIndexSetDenseSa(C.c_void_p(imgs[0]), dense) == awfm.AwFmSuccess
        assert awfm.parallel_search_locate(ix, lst, 6) == awfm.AwFmSuccess
        assert np.array_equal(lst.counts(), cnt)
        for i in range(0, 5003, 11):
            assert np.array_equal(lst.positions(i), pos[int(hit_off[i]):int(hit_off[i + 1])])
    lst.dealloc()
    ix.dealloc()


@pytest.mark.parametrize("kernel", [1, 2, 3, 4])  # AWFM_GPU_KERNEL_GROUP8 / 4 / 2 / 1 lanes per query
@pytest.mark.parametrize("alphabet_name", ["dna", "amino"])
def test_every_kernel_variant_is_bit_exact(oracle, awfm, require_gpu, wide, kernel, alphabet_name):
    amino = alphabet_name == "amino"
    letters = synth.AMINO_ALPHABET if amino else synth.DNA_ALPHABET
    alpha, oalpha = (awfm.AwFmAlphabetAmino, oracle.AMINO) if amino else (awfm.AwFmAlphabetDna, oracle.DNA)
    txt = synth.text(700 + kernel, 90000, letters).copy()
    txt[50:54] = ord("x")
    k = 3 if amino else 7
    ix = awfm.create_index(txt, alpha, 6, k)
    oi = oracle.Index.from_text(txt.tobytes(), oalpha, 6, k)
    chars, offsets = _mixed_queries(701, 5000, txt, letters, 1, 45, ambiguity=ord("z" if amino else "x"), upper=not amino)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    g = awfm.GpuIndex(ix)
    g.set_kernel(kernel)
    ranges, ho, p = g.locate_host(chars, offsets)
    assert np.array_equal(ranges[:, 0], sp) and np.array_equal(ranges[:, 1], ep)
    assert np.array_equal(ho, hit_off) and np.array_equal(p, pos)
    g.destroy()
    ix.dealloc()


def test_drop_in_aos_api_large_lists_are_packed_in_parallel(oracle, awfm, require_gpu):
    """lists beyond the serial threshold (4096 k-mers) with 8 host threads: mixed lengths with empty k-mers
    (CSR path) and a uniform-length list (fixed-length path); counts and every position list against the oracle"""
    txt = synth.text(91, 120000)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 4, 7)
    oi = oracle.Index.wrap(oracle.DNA, 4, 7, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                           ix.packed_sa())
    n = 20000
    chars, offsets = synth.mixed_queries(92, n, txt, synth.DNA_ALPHABET, 5, 33)
    mixed = [chars[int(offsets[i]):int(offsets[i + 1])].tobytes() for i in range(n)]
    for i in range(0, n, 997):
        mixed[i] = b""  # empty k-mers inside several chunks, and as the first entry
    uniform = [bytes(q) for q in np.concatenate([synth.random_queries(93, n // 2, 13), synth.planted_queries(94, n // 2, 13, txt)])]
    for kmers in (mixed, uniform):
        sp, ep, cnt, _ = oi.search_list(kmers)
        hit_off, pos, _ = oi.batch_locate(sp, ep)
        lst = awfm.KmerSearchList(n)
        lst.fill(kmers)
        awfm.parallel_search_count(ix, lst, 8)
        assert np.array_equal(lst.counts(), cnt)
        assert awfm.parallel_search_locate(ix, lst, 8) == awfm.AwFmSuccess
        assert np.array_equal(lst.counts(), cnt)
        for i in range(0, n, 13):
            assert np.array_equal(lst.positions(i), pos[int(hit_off[i]):int(hit_off[i + 1])]), f"k-mer {i}"
        lst.dealloc()
    ix.dealloc()


@pytest.mark.parametrize("alphabet_name", ["dna", "amino"])
def test_flat_locate_with_any_position_buffer_alignment_and_hit_count(oracle, awfm, require_gpu, wide, alphabet_name):
    """the walk kernel moves hits in 128-byte batches when the position buffer is 16-byte aligned and one by one
    otherwise; hit totals that are not a multiple of a batch exercise the tail of both paths"""
    import torch
    amino = alphabet_name == "amino"
    alphabet = synth.AMINO_ALPHABET if amino else synth.DNA_ALPHABET
    alpha, oalpha = (awfm.AwFmAlphabetAmino, oracle.AMINO) if amino else (awfm.AwFmAlphabetDna, oracle.DNA)
    k_seed, ratio = (2, 5) if amino else (6, 7)
    txt = synth.text(33, 90000, alphabet)
    ix = awfm.create_index(txt, alpha, ratio, k_seed)
    oi = oracle.Index.wrap(oalpha, ratio, k_seed, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                           ix.packed_sa())
    g = awfm.GpuIndex(ix)
    dev = torch.device("cuda")
    for count in (1, 7, 16, 17, 1000, 4097):
        K = 4 if amino else 9
        q = synth.planted_queries(300 + count, count, K, txt)
        chars, offsets = synth.fixed_csr(q)
        sp, ep, _, _ = oi.batch_search(chars, offsets)
        hit_off, pos, _ = oi.batch_locate(sp, ep)
        d_chars = torch.from_numpy(chars).to(dev)
        d_ranges = torch.zeros(count * 2, dtype=torch.int64, device=dev)
        g.search(d_chars.data_ptr(), 0, K, count, d_ranges.data_ptr(), 0)
        d_hit_off = torch.zeros(count + 1, dtype=torch.int64, device=dev)
        d_scratch = torch.zeros(awfm.GpuIndex.scan_scratch_bytes(count), dtype=torch.uint8, device=dev)
        total = g.hit_offsets(d_ranges.data_ptr(), count, d_hit_off.data_ptr(), d_scratch.data_ptr())
        assert total == len(pos)
        for shift in (0, 1):  # 16-byte aligned, then only 8-byte aligned
            d_pos = torch.zeros(total + 4, dtype=torch.int64, device=dev)
            g.locate(d_ranges.data_ptr(), d_hit_off.data_ptr(), count, total, d_pos.data_ptr() + 8 * shift)
            torch.cuda.synchronize()
            got = d_pos[shift:shift + total].cpu().numpy().view(np.uint64)
            assert np.array_equal(got, pos), f"{count} k-mers, buffer shift {shift}"
            assert int(d_pos[shift + total]) == 0  # nothing written past the end
    g.destroy()
    ix.dealloc()


def test_drop_in_aos_api_default_lanes(oracle, awfm, require_gpu, monkeypatch):
    """lists of 65536 k-mers and more are dealt to three host lanes on the one default image (no device list in
    the environment): counts and position lists against the oracle, twice in a row (staging buffers re-used)"""
    monkeypatch.delenv("AWFM_GPU_DEVICES", raising=False)
    txt = synth.text(95, 300000)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                           ix.packed_sa())
    n = 70001
    q = np.concatenate([synth.random_queries(96, n // 2, 15), synth.planted_queries(97, n - n // 2, 15, txt)])
    chars, offsets = synth.fixed_csr(q)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=4)
    hit_off, pos, _ = oi.batch_locate(sp, ep, threads=4)
    lst = awfm.KmerSearchList(n)
    lst.fill([bytes(r) for r in q])
    for _ in range(2):
        awfm.parallel_search_count(ix, lst, 8)
        assert np.array_equal(lst.counts(), cnt)
        assert awfm.parallel_search_locate(ix, lst, 8) == awfm.AwFmSuccess
        assert np.array_equal(lst.counts(), cnt)
        for i in list(range(0, n, 97)) + [n // 2 - 1, n // 2, n - 1]:
            assert np.array_equal(lst.positions(i), pos[int(hit_off[i]):int(hit_off[i + 1])]), f"k-mer {i}"
    from avxwindowfmindex_amd import _lib
    import ctypes as C
    imgs = (C.c_void_p * 4)()
    assert _lib.lib().awfmGpuIndexAcquireAll(ix.ptr, imgs, 4) == 3 and len({imgs[0], imgs[1], imgs[2]}) == 3
    lst.dealloc()
    ix.dealloc()


@pytest.mark.parametrize("n,seed_k,deep_k,K,pair", [(300000, 8, 12, 21, "1"), (300000, 8, 12, 13, "1"), (200000, 6, 9, 9, "1"),
                                                    (300000, 8, 12, 21, "0"), (250000, 12, 16, 24, "1"), (150000, 6, 9, 26, "1")])
def test_lookup_first_keeps_hits_bit_identical(oracle, awfm, require_gpu, wide, monkeypatch, n, seed_k, deep_k, K, pair):
    """"Lookup first" (encodeLookupKernel): the table entry of every k-mer is read while the batch is encoded, and only the
    k-mers that are still alive after it -- and the ones with ambiguity characters, which go to the general kernel -- are
    ordered and searched.  Forced on and off over the same batches (random + planted k-mers, ambiguity characters, upper
    case, every byte alignment): dense results, counts only, and the list of hits must all be the oracle's."""
    import torch
    monkeypatch.setenv("AWFM_GPU_PAIR", pair)
    txt = synth.text(n + 31, n, synth.DNA_ALPHABET).copy()
    txt[10:14] = ord("n")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, seed_k)
    oi = oracle.Index.wrap(oracle.DNA, 8, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                           ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    g.set_deep_seed(deep_k)
    Q = 70013
    q = np.concatenate([synth.random_queries(7, Q - Q // 8, K), synth.planted_queries(8, Q // 8, K, txt)]).copy()
    rng = np.random.default_rng(n + K)
    q = q[rng.permutation(Q)]
    flat = q.reshape(-1)
    flat[rng.random(flat.size) < 0.0005] = ord("x")
    up = rng.random(flat.size) < 0.3
    flat[up] = flat[up] & 0xDF
    chars, offsets = synth.fixed_csr(q)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    assert cnt.sum() > 0 and (cnt == 0).sum() > Q // 4
    dev = torch.device("cuda")
    for mode in ("1", "0"):
        monkeypatch.setenv("AWFM_GPU_LOOKUP_FIRST", mode)
        for mis in (0, 1, 2, 3):
            buf = torch.zeros(chars.size + 64, dtype=torch.uint8, device=dev)
            buf[mis:mis + chars.size] = torch.from_numpy(chars).to(dev)
            d_ranges = torch.full((Q * 2,), 7, dtype=torch.int64, device=dev)
            d_counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
            g.search_hits(buf.data_ptr() + mis, 0, K, Q, d_ranges.data_ptr(), d_counts.data_ptr())
            torch.cuda.synchronize()
            assert g.last_ordered_kernel_is_lookup() == (mode == "1")
            _check_hits_contract(d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2), d_counts.cpu().numpy().view(np.uint32), sp, ep, cnt)
        d_counts2 = torch.full((Q,), 7, dtype=torch.int32, device=dev)
        g.search_hits(buf.data_ptr() + 3, 0, K, Q, 0, d_counts2.data_ptr())  # counts only
        torch.cuda.synchronize()
        assert np.array_equal(d_counts2.cpu().numpy().view(np.uint32), cnt)
        # the list of the k-mers with hits
        cap = Q
        d_kmers = torch.zeros(cap, dtype=torch.int32, device=dev)
        d_hit_ranges = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
        d_num = torch.zeros(1, dtype=torch.int32, device=dev)
        g.search_hits_compact(buf.data_ptr() + 3, 0, K, Q, d_kmers.data_ptr(), d_hit_ranges.data_ptr(), cap, d_num.data_ptr())
        torch.cuda.synchronize()
        listed = int(d_num.item())
        assert listed == int((cnt > 0).sum())
        g.sort_hits(d_kmers.data_ptr(), d_hit_ranges.data_ptr(), listed)
        torch.cuda.synchronize()
        ids = d_kmers[:listed].cpu().numpy().view(np.uint32)
        r = d_hit_ranges[:listed * 2].cpu().numpy().view(np.uint64).reshape(listed, 2)
        assert np.array_equal(ids, np.flatnonzero(cnt > 0)) and np.array_equal(r[:, 0], sp[cnt > 0]) and np.array_equal(r[:, 1], ep[cnt > 0])
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("tail", [1, 31, 32, 63, 64, 255])
@pytest.mark.parametrize("K", [21, 12])
def test_lookup_first_all_hits_batch_whose_last_round_is_short(oracle, awfm, require_gpu, wide, monkeypatch, tail, K):
    """encodeLookupKernel reserves slots in blocks of 64 per wave; in the last partial round of the last share (nq % 256 in
    1..63) a block of 64 would reach past the share's region -- past the code array -- when every earlier k-mer was kept.
    Forced lookup first on batches in which EVERY k-mer has hits (nothing is dropped, no slack anywhere), with
    nq = 8 shares' worth of whole rounds + a short tail; K = 12 also runs the vector loads at the end of the batch (the
    caller's buffer ends with the batch: a guard page is not available, so the k-mers sit at the very end of an allocation)."""
    import torch
    monkeypatch.setenv("AWFM_GPU_LOOKUP_FIRST", "1")
    n = 200000
    txt = synth.text(n + 7, n, synth.DNA_ALPHABET).copy()
    seed_k, deep_k = (8, 12) if K == 21 else (6, 9)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, seed_k)
    oi = oracle.Index.wrap(oracle.DNA, 8, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    g.set_deep_seed(deep_k)
    dev = torch.device("cuda")
    for rounds in (8 * 64, 3):  # several tiles per share / a batch smaller than one share
        Q = rounds * 256 + tail
        q = synth.planted_queries(100 + tail, Q, K, txt)
        chars, offsets = synth.fixed_csr(q)
        sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=4)
        assert cnt.min() >= 1
        buf = torch.zeros(4096 + chars.size, dtype=torch.uint8, device=dev)
        buf[4096:] = torch.from_numpy(chars).to(dev)  # the batch ends where the buffer ends
        d_ranges = torch.full((Q * 2,), 7, dtype=torch.int64, device=dev)
        d_counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
        g.search_hits(buf.data_ptr() + 4096, 0, K, Q, d_ranges.data_ptr(), d_counts.data_ptr())
        torch.cuda.synchronize()
        assert g.last_ordered_kernel_is_lookup()
        assert g.last_ordered_kept() == Q
        _check_hits_contract(d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2), d_counts.cpu().numpy().view(np.uint32), sp, ep, cnt)
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("planted_share", [0.02, 0.6])
def test_list_sorted_scanned_and_located_without_a_host_wait(oracle, awfm, require_gpu, wide, planted_share):
    """awfmGpuSearchHitsCompact -> awfmGpuSortHitsOnDevice -> awfmGpuHitOffsetsOnDevice -> awfmGpuLocateOnDevice: the list's
    length and the number of hits are read on the device only.  The sorted list, its offsets and the positions must be what
    the host-counted calls (awfmGpuSortHits / awfmGpuHitOffsets / awfmGpuLocate) and the oracle give; a position buffer that
    is too small gets the first `capacity` hits and nothing behind them."""
    import torch
    n, K, Q = 300000, 15, 90011
    txt = synth.text(n + 3, n, synth.DNA_ALPHABET).copy()
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    g.set_deep_seed(11)
    m = int(Q * planted_share)
    q = np.concatenate([synth.random_queries(21, Q - m, K), synth.planted_queries(22, m, K, txt)])
    q = q[np.random.default_rng(5).permutation(Q)]
    chars, offsets = synth.fixed_csr(q)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=4)
    has = np.flatnonzero(cnt > 0)
    oho, opos, _ = oi.batch_locate(sp[has], ep[has], threads=4)
    dev = torch.device("cuda")
    d_chars = torch.from_numpy(chars).to(dev)
    cap = Q
    d_kmers = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_ranges = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
    d_num = torch.zeros(1, dtype=torch.int32, device=dev)
    d_off = torch.zeros(cap + 1, dtype=torch.int64, device=dev)
    d_scratch = torch.zeros(awfm.GpuIndex.scan_scratch_bytes(cap), dtype=torch.uint8, device=dev)
    for capacity_hits in (len(opos) + 100, max(len(opos) // 2, 1)):
        d_pos = torch.full((len(opos) + 200,), -1, dtype=torch.int64, device=dev)
        g.search_hits_compact(d_chars.data_ptr(), 0, K, Q, d_kmers.data_ptr(), d_ranges.data_ptr(), cap, d_num.data_ptr())
        g.sort_hits_on_device(d_kmers.data_ptr(), d_ranges.data_ptr(), cap, d_num.data_ptr(), Q)
        g.hit_offsets_on_device(0, d_ranges.data_ptr(), cap, d_off.data_ptr(), d_scratch.data_ptr())
        g.locate_on_device(d_ranges.data_ptr(), d_off.data_ptr(), cap, capacity_hits, d_pos.data_ptr())
        torch.cuda.synchronize()
        listed = int(d_num.item())
        assert listed == len(has)
        ids = d_kmers.cpu().numpy().view(np.uint32)
        r = d_ranges.cpu().numpy().view(np.uint64).reshape(cap, 2)
        assert np.array_equal(ids[:listed], has)
        assert np.array_equal(r[:listed, 0], sp[has]) and np.array_equal(r[:listed, 1], ep[has])
        assert np.all(ids[listed:] == 0xFFFFFFFF) and np.all(r[listed:, 0] > r[listed:, 1])
        off = d_off.cpu().numpy().view(np.uint64)
        assert np.array_equal(off[:listed + 1], oho) and np.all(off[listed:] == oho[-1])
        pos = d_pos.cpu().numpy()
        got = min(capacity_hits, len(opos))
        assert np.array_equal(pos[:got].view(np.uint64), opos[:got])
        assert np.all(pos[capacity_hits:] == -1), "written behind the capacity"
    g.destroy()
    ix.dealloc()


@pytest.mark.gpu
@pytest.mark.parametrize("dense_sa", [False, True])
@pytest.mark.parametrize("shape", ["sparse", "half", "clustered", "repeats"])
def test_list_tail_in_one_launch(oracle, awfm, require_gpu, wide, shape, dense_sa):
    """awfmGpuSearchHitsCompact -> awfmGpuListLocateOnDevice: the appended list comes out in k-mer order with its hit offsets
    and positions from ONE kernel (each workgroup finds its own prefix: no scan, no scratch), with and without the full suffix
    array, equal to the oracle's and to what the three calls it replaces leave; `clustered`: a stretch of 12 000 consecutive
    k-mers that all occur, i.e. more entries in one workgroup's range than its LDS slots hold (sub-ranges); a position buffer
    that is too small gets the first `capacity` hits and nothing behind them; offsets only when there is no buffer.  `repeats`:
    k-mers out of a tandem repeat with 5 * 10^3 hits each (expanded by the whole workgroup) and out of a 40-copy repeat (by a
    wave) among k-mers with a hit or two (by their lane)."""
    import torch
    n, K = 300000, 15
    Q = 90011 if shape != "clustered" else 3_000_017
    txt = synth.text(n + 3, n, synth.DNA_ALPHABET).copy()
    if shape == "repeats":
        txt[100000:110000] = np.frombuffer(b"ac" * 5000, np.uint8)
        txt[20000:22000] = np.tile(txt[20000:20050], 40)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    g.set_deep_seed(11)
    g.set_dense_sa(dense_sa)
    assert g.has_dense_sa == dense_sa
    if shape == "clustered":
        q = synth.random_queries(21, Q, K).copy()
        q[1_000_000:1_012_000] = synth.planted_queries(22, 12000, K, txt)
        q[Q - 300:] = synth.planted_queries(23, 300, K, txt)  # ... and the very last k-mers of the batch
    else:
        m = int(Q * (0.02 if shape in ("sparse", "repeats") else 0.6))
        q = np.concatenate([synth.random_queries(21, Q - m, K), synth.planted_queries(22, m, K, txt)])
        q = q[np.random.default_rng(5).permutation(Q)]
        if shape == "repeats":
            q[7::9001] = np.frombuffer((b"ac" * 8)[:K], np.uint8)
            q[4000::20011] = np.frombuffer((b"ca" * 8)[:K], np.uint8)
            q[11::1501] = txt[20010:20010 + K]
    chars, offsets = synth.fixed_csr(q)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=4)
    has = np.flatnonzero(cnt > 0)
    oho, opos, _ = oi.batch_locate(sp[has], ep[has], threads=4)
    dev = torch.device("cuda")
    d_chars = torch.from_numpy(chars).to(dev)
    cap = len(has) + len(has) // 3 + 17
    d_kmers = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_ranges = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
    d_num = torch.zeros(1, dtype=torch.int32, device=dev)
    d_skmers = torch.full((cap,), 5, dtype=torch.int32, device=dev)
    d_sranges = torch.full((cap * 2,), 5, dtype=torch.int64, device=dev)
    d_off = torch.full((cap + 1,), -3, dtype=torch.int64, device=dev)
    for capacity_hits in (len(opos) + 100, max(len(opos) // 2, 1), 0):
        d_pos = torch.full((len(opos) + 200,), -1, dtype=torch.int64, device=dev)
        g.search_hits_compact(d_chars.data_ptr(), 0, K, Q, d_kmers.data_ptr(), d_ranges.data_ptr(), cap, d_num.data_ptr())
        before = (d_kmers.clone(), d_ranges.clone())
        g.list_locate_on_device(d_kmers.data_ptr(), d_ranges.data_ptr(), cap, d_num.data_ptr(), Q, d_skmers.data_ptr(), d_sranges.data_ptr(),
                                d_off.data_ptr(), capacity_hits, d_pos.data_ptr() if capacity_hits else 0)
        torch.cuda.synchronize()
        assert torch.equal(before[0], d_kmers) and torch.equal(before[1], d_ranges), "the appended list was touched"
        listed = int(d_num.item())
        assert listed == len(has)
        ids = d_skmers.cpu().numpy().view(np.uint32)
        r = d_sranges.cpu().numpy().view(np.uint64).reshape(cap, 2)
        assert np.array_equal(ids[:listed], has)
        assert np.array_equal(r[:listed, 0], sp[has]) and np.array_equal(r[:listed, 1], ep[has])
        assert np.all(ids[listed:] == 0xFFFFFFFF) and np.all(r[listed:, 0] > r[listed:, 1])
        off = d_off.cpu().numpy().view(np.uint64)
        assert np.array_equal(off[:listed + 1], oho) and np.all(off[listed:] == oho[-1])
        pos = d_pos.cpu().numpy()
        got = min(capacity_hits, len(opos))
        assert np.array_equal(pos[:got].view(np.uint64), opos[:got])
        assert np.all(pos[capacity_hits:] == -1), "written behind the capacity"
    # a list that overflowed its capacity: the first `capacity` entries the search stored, in k-mer order
    small = max(len(has) // 3, 1)
    g.search_hits_compact(d_chars.data_ptr(), 0, K, Q, d_kmers.data_ptr(), d_ranges.data_ptr(), small, d_num.data_ptr())
    g.list_locate_on_device(d_kmers.data_ptr(), d_ranges.data_ptr(), small, d_num.data_ptr(), Q, d_skmers.data_ptr(), d_sranges.data_ptr(),
                            d_off.data_ptr(), 0, 0)
    torch.cuda.synchronize()
    assert int(d_num.item()) == len(has)
    stored = np.sort(d_kmers[:small].cpu().numpy().view(np.uint32))
    assert np.array_equal(d_skmers[:small].cpu().numpy().view(np.uint32), stored) and np.all(np.isin(stored, has))
    assert int(d_off[small].item()) == int(cnt[stored].sum())
    g.destroy()
    ix.dealloc()


@pytest.mark.gpu
def test_list_tail_of_a_long_list_takes_the_three_calls(oracle, awfm, require_gpu, wide, monkeypatch):
    """a list of more than 2^18 entries goes through copy + awfmGpuSortHitsOnDevice +
    awfmGpuHitOffsetsOnDevice + awfmGpuLocateOnDevice inside awfmGpuListLocateOnDevice: the same arrays come out"""
    import torch
    n, K, Q = 300000, 14, 700_001
    txt = synth.text(n + 5, n, synth.DNA_ALPHABET).copy()
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    q = synth.planted_queries(31, Q, K, txt).copy()
    q[::3] = synth.random_queries(32, len(q[::3]), K)
    chars, offsets = synth.fixed_csr(q)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=4)
    has = np.flatnonzero(cnt > 0)
    assert len(has) > (1 << 18)
    oho, opos, _ = oi.batch_locate(sp[has], ep[has], threads=4)
    dev = torch.device("cuda")
    d_chars = torch.from_numpy(chars).to(dev)
    for cap in (len(has) + 1000,):
        d_kmers = torch.zeros(cap, dtype=torch.int32, device=dev)
        d_ranges = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
        d_num = torch.zeros(1, dtype=torch.int32, device=dev)
        d_skmers = torch.zeros(cap, dtype=torch.int32, device=dev)
        d_sranges = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
        d_off = torch.zeros(cap + 1, dtype=torch.int64, device=dev)
        d_pos = torch.full((len(opos) + 8,), -1, dtype=torch.int64, device=dev)
        g.search_hits_compact(d_chars.data_ptr(), 0, K, Q, d_kmers.data_ptr(), d_ranges.data_ptr(), cap, d_num.data_ptr())
        g.list_locate_on_device(d_kmers.data_ptr(), d_ranges.data_ptr(), cap, d_num.data_ptr(), Q, d_skmers.data_ptr(), d_sranges.data_ptr(),
                                d_off.data_ptr(), len(opos) + 8, d_pos.data_ptr())
        torch.cuda.synchronize()
        listed = int(d_num.item())
        assert listed == len(has)
        assert np.array_equal(d_skmers[:listed].cpu().numpy().view(np.uint32), has)
        assert np.array_equal(d_off[:listed + 1].cpu().numpy().view(np.uint64), oho)
        assert np.array_equal(d_pos[:len(opos)].cpu().numpy().view(np.uint64), opos)
    g.destroy()
    ix.dealloc()


@pytest.mark.gpu
def test_lookup_prediction_launches_one_front_end_and_keeps_the_results(oracle, awfm, require_gpu, wide, monkeypatch):
    """Round 5: a sampled search publishes its sample's verdict in page-locked host memory, and a later search of the same
    k-mer length launches only the front end that verdict names (awfmGpuLastLookupFront: 0 both, 1 the lookup kernel alone
    with what it cannot finish left to the general kernel, 2 the ordering passes + ordered kernel alone).  Either front end
    alone must give every batch the oracle's counts and list -- the batch the prediction is wrong for included (k-mers drawn
    from the text right behind random ones and the other way round) -- and a wrong prediction must switch it off for a while."""
    import torch
    for name in ("AWFM_GPU_LOOKUP_FIRST", "AWFM_GPU_LOOKUP_PREDICT"):
        monkeypatch.delenv(name, raising=False)
    n, K, Q = 300000, 21, (1 << 20) + 5
    txt = synth.text(n + 41, n, synth.DNA_ALPHABET).copy()
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    g.set_deep_seed(12)
    dev = torch.device("cuda")
    batches = {}
    for name, q in (("random", synth.random_queries(9, Q, K).copy()), ("planted", synth.planted_queries(10, Q, K, txt).copy())):
        q[::1000, 3] = ord("n")  # some for the general kernel
        if name == "random":
            q[5::64] = synth.planted_queries(11, len(q[5::64]), K, txt)  # and some hits
        chars, offsets = synth.fixed_csr(q)
        sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=4)
        batches[name] = (torch.from_numpy(chars).to(dev), sp, ep, cnt)
    d_counts = torch.empty(Q, dtype=torch.int32, device=dev)
    cap = Q
    d_kmers = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_ranges = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
    d_num = torch.zeros(1, dtype=torch.int32, device=dev)

    def run(name, listed):
        d_chars, sp, ep, cnt = batches[name]
        if listed:
            g.search_hits_compact(d_chars.data_ptr(), 0, K, Q, d_kmers.data_ptr(), d_ranges.data_ptr(), cap, d_num.data_ptr())
            torch.cuda.synchronize()
            m = int(d_num.item())
            has = np.flatnonzero(cnt > 0)
            assert m == len(has), (name, m, len(has))
            ids = d_kmers[:m].cpu().numpy().view(np.uint32)
            order = np.argsort(ids)
            r = d_ranges[: 2 * m].cpu().numpy().view(np.uint64).reshape(m, 2)[order]
            assert np.array_equal(ids[order], has) and np.array_equal(r[:, 0], sp[has]) and np.array_equal(r[:, 1], ep[has]), name
        else:
            d_counts.fill_(7)
            g.search_hits(d_chars.data_ptr(), 0, K, Q, 0, d_counts.data_ptr())
            torch.cuda.synchronize()
            assert np.array_equal(d_counts.cpu().numpy().view(np.uint32), cnt), name
        return g.last_lookup_front()

    fronts = [run("random", False), run("random", True), run("random", False), run("random", True)]
    assert fronts[0] == 0 and fronts[2:] == [1, 1], fronts  # the first search has no verdict to go by; the third certainly has
    assert g.last_ordered_kernel_is_lookup()
    wrong = run("planted", False)  # predicted from the random batches: the lookup kernel alone, three quarters through the general kernel
    assert wrong == 1
    held = [run("planted", i % 2 == 1) for i in range(9)]
    assert held[:8] == [0] * 8, held  # the miss switched the prediction off for eight searches
    later = [run("planted", i % 2 == 0) for i in range(3)]
    assert later[-1] == 2, (held, later)  # ... and then the ordered kernels alone
    assert not g.last_ordered_kernel_is_lookup()
    assert run("random", True) == 2  # (wrong again, the other way round: slower, the same list)
    assert run("random", False) == 0
    monkeypatch.setenv("AWFM_GPU_LOOKUP_PREDICT", "0")
    assert [run("random", False), run("random", True)] == [0, 0]
    g.destroy()
    ix.dealloc()


def test_lookup_first_is_chosen_by_a_sample_of_the_batch(oracle, awfm, require_gpu, wide, monkeypatch):
    """Without $AWFM_GPU_LOOKUP_FIRST a batch of 2^20 k-mers or more is sampled (16384 k-mers at a fixed stride): random
    21-mers against a small text nearly all end at the deeper table -> encodeLookupKernel; k-mers drawn from the text all
    survive it -> the count + partition passes as before.  Counts against the oracle either way."""
    import torch
    monkeypatch.delenv("AWFM_GPU_LOOKUP_FIRST", raising=False)
    monkeypatch.setenv("AWFM_GPU_LOOKUP_PREDICT", "0")  # every search by its own sample (the prediction: test_lookup_prediction_*)
    n, K, Q = 300000, 21, (1 << 20) + 5
    txt = synth.text(n + 41, n, synth.DNA_ALPHABET).copy()
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    g.set_deep_seed(12)
    dev = torch.device("cuda")
    for name, q in (("random", synth.random_queries(9, Q, K).copy()), ("planted", synth.planted_queries(10, Q, K, txt).copy())):
        q[::1000, 3] = ord("n")  # some for the general kernel
        if name == "random":
            q[5::64] = synth.planted_queries(11, len(q[5::64]), K, txt)  # and some hits
        chars, offsets = synth.fixed_csr(q)
        _, _, cnt, _ = oi.batch_search(chars, offsets, threads=4)
        d_chars = torch.from_numpy(chars).to(dev)
        d_counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
        g.search_hits(d_chars.data_ptr(), 0, K, Q, 0, d_counts.data_ptr())
        torch.cuda.synchronize()
        assert g.last_ordered_kernel_is_lookup() == (name == "random"), name
        kept = g.last_ordered_kept()
        assert (kept < Q // 8) if name == "random" else (kept == Q), (name, kept)
        assert np.array_equal(d_counts.cpu().numpy().view(np.uint32), cnt), name
    g.destroy()
    ix.dealloc()


def _check_hits_contract(ranges, counts, sp, ep, cnt):
    """awfmGpuSearchHits: exact range and count for queries with hits, count 0 and an empty range otherwise"""
    hit = cnt > 0
    assert np.array_equal(counts, cnt), "counts differ"
    assert np.array_equal(ranges[hit, 0], sp[hit]) and np.array_equal(ranges[hit, 1], ep[hit]), "ranges of hits differ"
    assert np.all(ranges[~hit, 0] > ranges[~hit, 1]), "a query without hits must have an empty range"


@pytest.mark.parametrize("lookup_first", [None, "0", "1"])
def test_counts_of_a_dense_hit_batch_come_home_from_search_order(oracle, awfm, require_gpu, wide, monkeypatch, lookup_first):
    """counts only through the seed-order search (round 6, awfm_count_order_kernel.h): orderedSearchKernel leaves {k-mer number,
    count} in search order, countScatterKernel / countPlaceKernel take them to counts[number] -- 700 001 k-mers are 22
    buckets of k-mer numbers (2^15 each), the last one partly filled; k-mers with ambiguity characters are the general
    kernel's; absent k-mers get their 0 from the same passes.  With the lookup kernel in front ($AWFM_GPU_LOOKUP_FIRST=1) most
    k-mers' counts are stored by THAT kernel and only what it leaves goes through the order: the passes must lay their records
    over what is there.  Against the oracle, from ASCII and from packed k-mers."""
    import torch
    if lookup_first is None:
        monkeypatch.delenv("AWFM_GPU_LOOKUP_FIRST", raising=False)
    else:
        monkeypatch.setenv("AWFM_GPU_LOOKUP_FIRST", lookup_first)
    n, K, Q = 400000, 21, 700001
    txt = synth.text(4242, n, synth.DNA_ALPHABET).copy()
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    g.set_deep_seed(10)
    q = np.concatenate([synth.planted_queries(61, Q - Q // 8, K, txt), synth.random_queries(62, Q // 8, K)]).copy()
    rng = np.random.default_rng(7)
    q = q[rng.permutation(Q)]
    flat = q.reshape(-1)
    flat[rng.random(flat.size) < 0.0005] = ord("n")
    chars, offsets = synth.fixed_csr(q)
    _, _, cnt, _ = oi.batch_search(chars, offsets, threads=os.cpu_count() or 1)
    assert (cnt > 0).sum() > Q // 2 and (cnt == 0).sum() > Q // 16
    dev = torch.device("cuda")
    d_chars = torch.from_numpy(chars).to(dev)
    for _ in range(2):  # (the second search finds the scratch and its counters as the first left them)
        d_counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
        g.search_hits(d_chars.data_ptr(), 0, K, Q, 0, d_counts.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d_counts.cpu().numpy().view(np.uint32), cnt)
    clean = ~(q == ord("n")).any(axis=1)
    packed = awfm.pack_kmers(np.ascontiguousarray(q[clean]), awfm.AwFmAlphabetDna)
    d_packed = torch.from_numpy(packed.view(np.int64)).to(dev)
    m = int(clean.sum())
    d_counts = torch.full((m,), 7, dtype=torch.int32, device=dev)
    g.search_hits_packed(d_packed.data_ptr(), K, m, 0, d_counts.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_counts.cpu().numpy().view(np.uint32), cnt[clean])
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("n,ratio,seed_k,deep_k,K", [(300000, 8, 8, 0, 21), (300000, 5, 8, 0, 8), (200000, 8, 6, 9, 32),
                                                     (200000, 8, 6, 9, 7), (4096, 3, 4, 0, 13), (100000, 8, 1, 0, 5),
                                                     (150000, 8, 10, 11, 11), (300000, 8, 12, 16, 21)])
def test_ordered_hits_only_search_is_exact_on_hits(oracle, awfm, require_gpu, wide, n, ratio, seed_k, deep_k, K):
    """awfmGpuSearchHits with the ordered path forced on (fixed-length DNA batches): ambiguity characters and upper
    case included, query buffer at every byte alignment, ranges only / counts only / both, then the locate
    pipeline on top of the hits-only ranges"""
    import torch
    txt = synth.text(n + 13, n, synth.DNA_ALPHABET).copy()
    txt[10:14] = ord("n")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, ratio, seed_k)
    oi = oracle.Index.wrap(oracle.DNA, ratio, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                           ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    if deep_k:
        g.set_deep_seed(deep_k)
    Q = 30011
    q = np.concatenate([synth.random_queries(5, Q // 2, K), synth.planted_queries(6, Q - Q // 2, K, txt)]).copy()
    rng = np.random.default_rng(n + K)
    flat = q.reshape(-1)
    flat[rng.random(flat.size) < 0.002] = ord("x")      # ambiguity characters: those k-mers go to the general kernel
    up = rng.random(flat.size) < 0.3
    flat[up] = flat[up] & 0xDF
    chars, offsets = synth.fixed_csr(q)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    assert cnt.sum() > 0
    dev = torch.device("cuda")
    for mis in range(4):
        buf = torch.zeros(chars.size + 64, dtype=torch.uint8, device=dev)
        buf[mis:mis + chars.size] = torch.from_numpy(chars).to(dev)
        d_ranges = torch.full((Q * 2,), 7, dtype=torch.int64, device=dev)
        d_counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
        g.search_hits(buf.data_ptr() + mis, 0, K, Q, d_ranges.data_ptr(), d_counts.data_ptr())
        torch.cuda.synchronize()
        r = d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2)
        _check_hits_contract(r, d_counts.cpu().numpy().view(np.uint32), sp, ep, cnt)
    d_ranges2 = torch.full((Q * 2,), 7, dtype=torch.int64, device=dev)
    g.search_hits(buf.data_ptr() + 3, 0, K, Q, d_ranges2.data_ptr(), 0)      # ranges only
    d_counts2 = torch.full((Q,), 7, dtype=torch.int32, device=dev)
    g.search_hits(buf.data_ptr() + 3, 0, K, Q, 0, d_counts2.data_ptr())      # counts only
    torch.cuda.synchronize()
    assert torch.equal(d_ranges2, d_ranges) and torch.equal(d_counts2, d_counts)
    # locate on top of the hits-only ranges
    d_hit_off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
    d_scratch = torch.zeros(awfm.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
    total = g.hit_offsets(d_ranges.data_ptr(), Q, d_hit_off.data_ptr(), d_scratch.data_ptr())
    assert total == len(pos) and np.array_equal(d_hit_off.cpu().numpy().view(np.uint64), hit_off)
    d_hit_off2 = torch.zeros(Q + 1, dtype=torch.int64, device=dev)  # the same offsets scanned from the 32-bit counts
    assert g.hit_offsets_from_counts(d_counts.data_ptr(), Q, d_hit_off2.data_ptr(), d_scratch.data_ptr()) == total
    assert torch.equal(d_hit_off2, d_hit_off)
    d_pos = torch.zeros(max(total, 1), dtype=torch.int64, device=dev)
    g.locate(d_ranges.data_ptr(), d_hit_off.data_ptr(), Q, total, d_pos.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_pos[:total].cpu().numpy().view(np.uint64), pos)
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("shape", ["one-kmer", "two-buckets", "tiny", "every-bucket-small", "sorted"])
def test_bucketed_order_with_skewed_and_tiny_batches(oracle, awfm, require_gpu, wide, shape):
    """the partitioned order leaves the bucket of a record to its position: batches in which one bucket holds everything,
    in which most buckets are empty, and in which a chunk of 16 records crosses many buckets"""
    import torch
    n, K, seed_k = 400_000, 19, 8
    txt = synth.text(771, n)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, seed_k)
    oi = oracle.Index.wrap(oracle.DNA, 8, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    planted = synth.planted_queries(772, 40_000, K, txt)
    if shape == "one-kmer":
        q = np.repeat(planted[:1], 20_000, axis=0)
    elif shape == "two-buckets":
        q = np.concatenate([np.repeat(planted[:1], 9_000, axis=0), np.repeat(planted[1:2], 11_111, axis=0), planted[:3]])
    elif shape == "tiny":
        q = np.concatenate([planted[:37], synth.random_queries(773, 40, K)])
    elif shape == "every-bucket-small":
        q = np.concatenate([planted[:3000], synth.random_queries(774, 3000, K)])  # 6000 k-mers over 2048 buckets
    else:
        q = planted[np.lexsort(planted[:, ::-1].T)]  # the batch arrives sorted: long runs of one bucket after another
    chars, offsets = synth.fixed_csr(q)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    Q = len(q)
    d_chars = torch.from_numpy(chars.copy()).cuda()
    d_ranges = torch.full((Q * 2,), 7, dtype=torch.int64, device="cuda")
    d_counts = torch.full((Q,), 7, dtype=torch.int32, device="cuda")
    g.search_hits(d_chars.data_ptr(), 0, K, Q, d_ranges.data_ptr(), d_counts.data_ptr())
    torch.cuda.synchronize()
    _check_hits_contract(d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2), d_counts.cpu().numpy().view(np.uint32), sp, ep, cnt)
    # the same k-mers bit-packed (the packed words are the partition's code array: no encoding pass)
    d_packed = torch.from_numpy(awfm.pack_kmers(q).view(np.int64)).cuda()
    d_ranges.fill_(7)
    d_counts.fill_(7)
    g.search_hits_packed(d_packed.data_ptr(), K, Q, d_ranges.data_ptr(), d_counts.data_ptr())
    torch.cuda.synchronize()
    _check_hits_contract(d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2), d_counts.cpu().numpy().view(np.uint32), sp, ep, cnt)
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("K,packed", [(19, False), (19, True), (32, False)])
def test_sparse_hit_list_matches_the_dense_results(oracle, awfm, require_gpu, wide, K, packed):
    """awfmGpuSearchHitsCompact + awfmGpuSortHits: the k-mers with hits as a list in k-mer order (ambiguous k-mers, which
    the general kernel answers at the end of the call, included), the same list out of dense results
    (awfmGpuCompactHits), locate on top of the list, and a list that overflows its capacity says so"""
    import torch
    n, seed_k, Q = 300_000, 8, 20_000
    txt = synth.text(881, n).copy()
    txt[5000:5040] = ord("n")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, seed_k)
    oi = oracle.Index.wrap(oracle.DNA, 8, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    q = np.concatenate([synth.random_queries(882, Q - 300, K), synth.planted_queries(883, 300, K, synth.text(881, n))]).copy()
    q = q[np.random.default_rng(5).permutation(Q)]
    if not packed:
        q[17, 3] = ord("n")      # no hit
        q[18] = ord("n")         # all n: matches inside the run of the text -- through the general kernel
    chars, offsets = synth.fixed_csr(q)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    ho, pos, _ = oi.batch_locate(sp, ep)
    hits = np.flatnonzero(cnt)
    assert 290 <= hits.size < 400 and (packed or cnt[18] > 0)  # (a planted k-mer may overlap the run of n)
    dev = torch.device("cuda")
    d_in = (torch.from_numpy(awfm.pack_kmers(q).view(np.int64)) if packed else torch.from_numpy(chars.copy())).to(dev)
    cap = 1024
    d_kmers = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_ranges = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
    d_num = torch.zeros(1, dtype=torch.int32, device=dev)
    g.search_hits_compact(d_in.data_ptr(), 0, K, Q, d_kmers.data_ptr(), d_ranges.data_ptr(), cap, d_num.data_ptr(), packed=packed)
    g.sort_hits(d_kmers.data_ptr(), d_ranges.data_ptr(), cap)
    torch.cuda.synchronize()
    assert int(d_num.item()) == hits.size
    kmers = d_kmers.cpu().numpy().view(np.uint32)
    ranges = d_ranges.cpu().numpy().view(np.uint64).reshape(cap, 2)
    assert np.array_equal(kmers[:hits.size], hits) and np.all(kmers[hits.size:] == 0xFFFFFFFF)
    assert np.array_equal(ranges[:hits.size, 0], sp[hits]) and np.array_equal(ranges[:hits.size, 1], ep[hits])
    assert np.all(ranges[hits.size:, 0] > ranges[hits.size:, 1])
    # locate over the list as if it were the batch
    d_off = torch.zeros(cap + 1, dtype=torch.int64, device=dev)
    d_scratch = torch.zeros(awfm.GpuIndex.scan_scratch_bytes(cap), dtype=torch.uint8, device=dev)
    total = g.hit_offsets(d_ranges.data_ptr(), cap, d_off.data_ptr(), d_scratch.data_ptr())
    assert total == len(pos)
    d_pos = torch.zeros(max(total, 1), dtype=torch.int64, device=dev)
    g.locate(d_ranges.data_ptr(), d_off.data_ptr(), cap, total, d_pos.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_pos[:total].cpu().numpy().view(np.uint64), pos)  # lists in k-mer order = the flat list of the batch
    # a capacity below the number of hits: the count still says how many there are
    g.search_hits_compact(d_in.data_ptr(), 0, K, Q, d_kmers.data_ptr(), d_ranges.data_ptr(), 100, d_num.data_ptr(), packed=packed)
    torch.cuda.synchronize()
    assert int(d_num.item()) == hits.size
    # the same list out of dense results
    if not packed:
        d_dr = torch.zeros(Q * 2, dtype=torch.int64, device=dev)
        d_dc = torch.zeros(Q, dtype=torch.int32, device=dev)
        g.search_hits(d_in.data_ptr(), 0, K, Q, d_dr.data_ptr(), d_dc.data_ptr())
        d_flags = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
        d_scratch2 = torch.zeros(awfm.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
        d_k2 = torch.zeros(cap, dtype=torch.int32, device=dev)
        d_r2 = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
        g.compact_hits(d_dc.data_ptr(), d_dr.data_ptr(), Q, d_flags.data_ptr(), d_scratch2.data_ptr(), d_k2.data_ptr(),
                       d_r2.data_ptr(), cap, d_num.data_ptr())
        torch.cuda.synchronize()
        assert int(d_num.item()) == hits.size
        assert np.array_equal(d_k2.cpu().numpy().view(np.uint32), kmers)
        assert np.array_equal(d_r2.cpu().numpy().view(np.uint64).reshape(cap, 2)[:hits.size], ranges[:hits.size])
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("K,csr", [(19, False), (32, False), (0, True)])
def test_results_in_search_order_are_a_permutation_with_exact_ranges(oracle, awfm, require_gpu, wide, K, csr):
    """awfmGpuSearchHitsInOrder: every k-mer exactly once, {number, range} in the order the seed-order search took them
    (k-mers with ambiguity characters, answered by the general kernel, at the end); locate over that order gives every
    k-mer's list, in BWT order, under its number"""
    import torch
    n, seed_k, Q = 300_000, 8, 20_001
    txt = synth.text(891, n).copy()
    txt[7000:7050] = ord("n")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, seed_k)
    oi = oracle.Index.wrap(oracle.DNA, 8, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    if csr:
        chars, offsets = _mixed_queries(892, Q, synth.text(891, n), synth.DNA_ALPHABET, 1, 40, ambiguity=ord("x"), upper=True)
    else:
        q = np.concatenate([synth.random_queries(893, Q // 3, K), synth.planted_queries(894, Q - Q // 3, K, synth.text(891, n))]).copy()
        q = q[np.random.default_rng(6).permutation(Q)]
        q[5, 1] = ord("n")
        q[6] = ord("n")
        chars, offsets = synth.fixed_csr(q)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    ho, pos, _ = oi.batch_locate(sp, ep)
    dev = torch.device("cuda")
    d_chars = torch.from_numpy(chars.copy()).to(dev)
    d_off = torch.from_numpy(offsets.view(np.int64).copy()).to(dev) if csr else None
    d_kmers = torch.full((Q,), -1, dtype=torch.int32, device=dev)
    d_ranges = torch.full((Q * 2,), 7, dtype=torch.int64, device=dev)
    # (round 6: with the 32-bit counts in the same order -- awfmGpuSearchHitsInOrderCounts --, from which the hit offsets scan)
    d_ocounts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
    g.search_hits_in_order(d_chars.data_ptr(), d_off.data_ptr() if csr else 0, K, Q, d_kmers.data_ptr(), d_ranges.data_ptr(),
                           d_order_counts=d_ocounts.data_ptr())
    d_hoff = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
    d_scratch = torch.zeros(awfm.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
    total = g.hit_offsets(d_ranges.data_ptr(), Q, d_hoff.data_ptr(), d_scratch.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_ocounts.cpu().numpy().view(np.uint32), cnt[d_kmers.cpu().numpy().astype(np.int64)]), "counts in search order"
    d_hoff2 = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
    assert g.hit_offsets_from_counts(d_ocounts.data_ptr(), Q, d_hoff2.data_ptr(), d_scratch.data_ptr()) == total and torch.equal(d_hoff2, d_hoff)
    d_pos = torch.zeros(max(total, 1), dtype=torch.int64, device=dev)
    g.locate(d_ranges.data_ptr(), d_hoff.data_ptr(), Q, total, d_pos.data_ptr())
    torch.cuda.synchronize()
    kmers = d_kmers.cpu().numpy().astype(np.int64)
    assert np.array_equal(np.sort(kmers), np.arange(Q)), "not a permutation of the batch"
    r = d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2)
    hit = cnt[kmers] > 0
    assert np.array_equal(r[hit, 0], sp[kmers][hit]) and np.array_equal(r[hit, 1], ep[kmers][hit])
    assert np.all(r[~hit, 0] > r[~hit, 1])
    assert total == len(pos)
    off_o = d_hoff.cpu().numpy().view(np.uint64)
    pos_o = d_pos[:total].cpu().numpy().view(np.uint64)
    assert np.array_equal(np.diff(off_o), cnt[kmers].astype(np.uint64))
    for e in list(range(0, Q, 37)) + [Q - 1]:
        i = kmers[e]
        assert np.array_equal(pos_o[int(off_o[e]):int(off_o[e + 1])], pos[int(ho[i]):int(ho[i + 1])]), (e, i)
    g.destroy()
    ix.dealloc()


def test_hits_only_search_falls_back_to_the_general_kernel(oracle, awfm, require_gpu):
    """batches the ordered path does not cover (CSR offsets, k-mers shorter than the seed or longer than 32
    characters, amino indices) still honour the hits-only contract"""
    import torch
    dev = torch.device("cuda")
    txt = synth.text(17, 150000)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                           ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    chars, offsets = synth.mixed_queries(18, 9000, txt, synth.DNA_ALPHABET, 1, 40)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    d_chars, d_off = torch.from_numpy(chars).to(dev), torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_ranges = torch.zeros(9000 * 2, dtype=torch.int64, device=dev)
    d_counts = torch.zeros(9000, dtype=torch.int32, device=dev)
    g.search_hits(d_chars.data_ptr(), d_off.data_ptr(), 0, 9000, d_ranges.data_ptr(), d_counts.data_ptr())
    torch.cuda.synchronize()
    _check_hits_contract(d_ranges.cpu().numpy().view(np.uint64).reshape(-1, 2), d_counts.cpu().numpy().view(np.uint32),
                         sp, ep, cnt)
    for K in (5, 36):
        q = np.concatenate([synth.random_queries(19, 2000, K), synth.planted_queries(20, 2000, K, txt)])
        c, o = synth.fixed_csr(q)
        sp, ep, cnt, _ = oi.batch_search(c, o)
        d_c = torch.from_numpy(c).to(dev)
        d_ranges = torch.zeros(4000 * 2, dtype=torch.int64, device=dev)
        d_counts = torch.zeros(4000, dtype=torch.int32, device=dev)
        g.search_hits(d_c.data_ptr(), 0, K, 4000, d_ranges.data_ptr(), d_counts.data_ptr())
        torch.cuda.synchronize()
        _check_hits_contract(d_ranges.cpu().numpy().view(np.uint64).reshape(-1, 2),
                             d_counts.cpu().numpy().view(np.uint32), sp, ep, cnt)
    g.destroy()
    ix.dealloc()
    atxt = synth.text(21, 60000, synth.AMINO_ALPHABET)
    aix = awfm.create_index(atxt, awfm.AwFmAlphabetAmino, 8, 3)
    aoi = oracle.Index.wrap(oracle.AMINO, 8, 3, aix.bwt_length, aix.blocks(), aix.prefix_sums(), aix.seed_table(),
                            aix.packed_sa())
    ag = awfm.GpuIndex(aix)
    ag.set_ordered(1)
    q = np.concatenate([synth.random_queries(22, 2000, 6, synth.AMINO_ALPHABET), synth.planted_queries(23, 2000, 6, atxt)])
    c, o = synth.fixed_csr(q)
    sp, ep, cnt, _ = aoi.batch_search(c, o)
    d_c = torch.from_numpy(c).to(dev)
    d_ranges = torch.zeros(4000 * 2, dtype=torch.int64, device=dev)
    d_counts = torch.zeros(4000, dtype=torch.int32, device=dev)
    ag.search_hits(d_c.data_ptr(), 0, 6, 4000, d_ranges.data_ptr(), d_counts.data_ptr())
    torch.cuda.synchronize()
    _check_hits_contract(d_ranges.cpu().numpy().view(np.uint64).reshape(-1, 2), d_counts.cpu().numpy().view(np.uint32),
                         sp, ep, cnt)
    ag.destroy()
    aix.dealloc()


def test_drop_in_aos_api_uses_the_ordered_search_when_forced(oracle, awfm, require_gpu, monkeypatch):
    """AWFM_GPU_ORDERED=1: the AoS entry points (which only report hits) search uniform-length lists in seed order"""
    monkeypatch.setenv("AWFM_GPU_ORDERED", "1")
    txt = synth.text(98, 200000)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                           ix.packed_sa())
    n = 30000
    q = np.concatenate([synth.random_queries(99, n // 2, 17), synth.planted_queries(100, n - n // 2, 17, txt)]).copy()
    q[7, 5] = ord("N")
    chars, offsets = synth.fixed_csr(q)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    lst = awfm.KmerSearchList(n)
    lst.fill([bytes(r) for r in q])
    awfm.parallel_search_count(ix, lst, 4)
    assert np.array_equal(lst.counts(), cnt)
    assert awfm.parallel_search_locate(ix, lst, 4) == awfm.AwFmSuccess
    assert np.array_equal(lst.counts(), cnt)
    for i in range(0, n, 29):
        assert np.array_equal(lst.positions(i), pos[int(hit_off[i]):int(hit_off[i + 1])]), f"k-mer {i}"
    lst.dealloc()
    ix.dealloc()


@pytest.mark.parametrize("n,ratio,seed_k,deep_k", [(300000, 8, 8, 0), (200000, 8, 6, 9), (4096, 3, 4, 0), (100000, 8, 1, 0),
                                                   (150000, 8, 10, 11), (250000, 7, 12, 0), (250000, 8, 12, 16)])
def test_ordered_hits_only_search_of_mixed_length_batches(oracle, awfm, require_gpu, wide, n, ratio, seed_k, deep_k):
    """CSR batches through the ordered path (forced on): lengths 0..40, so k-mers start from the deeper table, the
    seed table or a letter range (shorter than the seed), and empty / over-long / ambiguous ones go to the general
    kernel; counts, hit ranges, hit offsets and positions against the oracle"""
    import torch
    txt = synth.text(n + 17, n, synth.DNA_ALPHABET).copy()
    txt[20:24] = ord("n")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, ratio, seed_k)
    oi = oracle.Index.wrap(oracle.DNA, ratio, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                           ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    if deep_k:
        g.set_deep_seed(deep_k)
    Q = 25013
    chars, offsets = _mixed_queries(4000 + n, Q, txt, synth.DNA_ALPHABET, 0, min(40, n), ambiguity=ord("x"), upper=True)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    dev = torch.device("cuda")
    d_chars = torch.from_numpy(np.concatenate([chars, np.zeros(8, np.uint8)])).to(dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_ranges = torch.full((Q * 2,), 7, dtype=torch.int64, device=dev)
    d_counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
    assert g.search_hits_is_ordered(True, 0, Q)
    g.search_hits(d_chars.data_ptr(), d_off.data_ptr(), 0, Q, d_ranges.data_ptr(), d_counts.data_ptr())
    torch.cuda.synchronize()
    _check_hits_contract(d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2), d_counts.cpu().numpy().view(np.uint32),
                         sp, ep, cnt)
    d_hit_off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
    d_scratch = torch.zeros(awfm.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
    total = g.hit_offsets(d_ranges.data_ptr(), Q, d_hit_off.data_ptr(), d_scratch.data_ptr())
    assert total == len(pos) and np.array_equal(d_hit_off.cpu().numpy().view(np.uint64), hit_off)
    d_pos = torch.zeros(max(total, 1), dtype=torch.int64, device=dev)
    g.locate(d_ranges.data_ptr(), d_hit_off.data_ptr(), Q, total, d_pos.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_pos[:total].cpu().numpy().view(np.uint64), pos)
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("n,seed_k,deep_k,pair,lo,hi", [(300000, 8, 12, "1", 0, 40), (300000, 8, 12, "0", 1, 32), (200000, 6, 9, "1", 0, 36),
                                                        (4096, 3, 5, "1", 0, 20), (250000, 12, 16, "1", 8, 30), (150000, 1, 2, "1", 0, 12)])
def test_mixed_length_lookup_first_keeps_hits_bit_identical(oracle, awfm, require_gpu, wide, monkeypatch, n, seed_k, deep_k, pair, lo, hi):
    """"Lookup first" for mixed-length batches (mixedLookupSearchKernel): one table entry per k-mer -- the table of its own
    length when it is shorter than the deeper table's k-mers, the deeper table otherwise --, the survivors stepped by the
    kernel that looked them up, what it does not cover (ambiguity characters, no characters, more than 32) left to the
    general kernel.  Forced on and off over the same batch (random + planted k-mers of every length, ambiguity characters,
    upper case, every byte alignment): dense results, counts only and the list of hits must all be the oracle's."""
    import torch
    monkeypatch.setenv("AWFM_GPU_PAIR", pair)
    txt = synth.text(n + 57, n, synth.DNA_ALPHABET).copy()
    txt[10:14] = ord("n")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, seed_k)
    oi = oracle.Index.wrap(oracle.DNA, 8, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    g.set_deep_seed(deep_k)
    Q = 70013
    chars, offsets = _mixed_queries(5000 + n, Q, txt, synth.DNA_ALPHABET, lo, min(hi, n), ambiguity=ord("x"), upper=True)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    assert cnt.sum() > 0 and (cnt == 0).sum() > Q // 16
    dev = torch.device("cuda")
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    for mode in ("1", "0"):
        monkeypatch.setenv("AWFM_GPU_MIXED_LOOKUP", mode)
        for mis in (0, 1, 2, 3):
            buf = torch.zeros(chars.size + 4, dtype=torch.uint8, device=dev)  # (the batch all but ends where the buffer ends)
            buf[mis:mis + chars.size] = torch.from_numpy(chars).to(dev)
            d_ranges = torch.full((Q * 2,), 7, dtype=torch.int64, device=dev)
            d_counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
            g.search_hits(buf.data_ptr() + mis, d_off.data_ptr(), 0, Q, d_ranges.data_ptr(), d_counts.data_ptr())
            torch.cuda.synchronize()
            assert g.last_ordered_kernel_is_lookup() == (mode == "1")
            _check_hits_contract(d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2), d_counts.cpu().numpy().view(np.uint32), sp, ep, cnt)
        d_counts2 = torch.full((Q,), 7, dtype=torch.int32, device=dev)
        g.search_hits(buf.data_ptr() + 3, d_off.data_ptr(), 0, Q, 0, d_counts2.data_ptr())  # counts only
        torch.cuda.synchronize()
        assert np.array_equal(d_counts2.cpu().numpy().view(np.uint32), cnt)
        # the list of the k-mers with hits
        cap = Q
        d_kmers = torch.zeros(cap, dtype=torch.int32, device=dev)
        d_hit_ranges = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
        d_num = torch.zeros(1, dtype=torch.int32, device=dev)
        g.search_hits_compact(buf.data_ptr() + 3, d_off.data_ptr(), 0, Q, d_kmers.data_ptr(), d_hit_ranges.data_ptr(), cap, d_num.data_ptr())
        torch.cuda.synchronize()
        listed = int(d_num.item())
        assert listed == int((cnt > 0).sum())
        g.sort_hits(d_kmers.data_ptr(), d_hit_ranges.data_ptr(), listed)
        torch.cuda.synchronize()
        ids = d_kmers[:listed].cpu().numpy().view(np.uint32)
        r = d_hit_ranges[:listed * 2].cpu().numpy().view(np.uint64).reshape(listed, 2)
        assert np.array_equal(ids, np.flatnonzero(cnt > 0)) and np.array_equal(r[:, 0], sp[cnt > 0]) and np.array_equal(r[:, 1], ep[cnt > 0])
    # the instrumented pass over the same k-mers (what bench.py prices the kernel by) reads them the same way
    tally = g.mixed_lookup_line_tally(buf.data_ptr() + 3, d_off.data_ptr(), Q)
    lens = np.diff(offsets.astype(np.int64))
    is_letter = np.isin(chars, np.frombuffer(b"acgtuACGTU", np.uint8))
    bad = np.add.reduceat(~is_letter, offsets[:-1].astype(np.int64).clip(max=max(chars.size - 1, 0))) * (lens > 0) > 0 if chars.size else np.zeros(Q, bool)
    general = (lens == 0) | (lens > 32) | bad
    assert tally["general_kmers"] == int(general.sum())
    assert tally["pair_level_lines"] + tally["nuc_level_lines"] <= tally["block_reads_executed"] <= 2 * 17 * tally["kmers_alive_after_the_table"]
    assert tally["kmers_with_hits"] == int(((cnt > 0) & ~general).sum())
    assert tally["kmers_alive_after_the_table"] <= int(((lens > deep_k) & ~general).sum())
    assert 0 < tally["length_table_lines"] + tally["deep_table_lines"] <= int((~general).sum())
    assert g.length_tables[0] == 8 * (4 ** deep_k - 4) // 3
    g.destroy()
    ix.dealloc()


@pytest.mark.parametrize("seed_order", [True, False])
def test_mixed_length_lookup_first_is_chosen_by_a_sample_of_the_batch(oracle, awfm, require_gpu, wide, monkeypatch, seed_order):
    """Without $AWFM_GPU_MIXED_LOOKUP a mixed-length batch of 2^20 k-mers or more is sampled: random 8..30-mers against a
    small text mostly end at their table entry -> mixedLookupSearchKernel; k-mers drawn from the text, most of them longer
    than the deeper table's, survive it -> the 16-byte-record path as before.  Counts against the oracle either way.
    seed_order False: the batch is below the size from which the seed-order path applies (not forced on here) -- there is
    no other front end to choose, the lookup kernel takes both batches."""
    import torch
    monkeypatch.delenv("AWFM_GPU_MIXED_LOOKUP", raising=False)
    monkeypatch.setenv("AWFM_GPU_LOOKUP_PREDICT", "0")  # every search by its own sample (the prediction: the next test)
    n, Q = 300000, (1 << 20) + 77
    txt = synth.text(n + 43, n, synth.DNA_ALPHABET).copy()
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    if seed_order:
        g.set_ordered(1)
    g.set_deep_seed(12)
    assert g.search_hits_is_ordered(True, 0, Q) == seed_order
    dev = torch.device("cuda")
    rng = np.random.default_rng(5)
    lengths = rng.integers(8, 31, Q)
    offsets = np.zeros(Q + 1, np.uint64)
    np.cumsum(lengths, out=offsets[1:])
    total = int(offsets[-1])
    starts = rng.integers(0, n - 40, Q)
    for name in ("random", "planted"):
        if name == "random":
            chars = np.frombuffer(synth.DNA_ALPHABET, np.uint8)[rng.integers(0, 4, total)].copy()
        else:
            idx = np.repeat(starts, lengths) + (np.arange(total) - np.repeat(offsets[:-1].astype(np.int64), lengths))
            chars = txt[idx].copy()
        chars[::5000] = ord("n")  # some for the general kernel
        _, _, cnt, _ = oi.batch_search(chars, offsets, threads=4)
        d_chars = torch.from_numpy(np.concatenate([chars, np.zeros(8, np.uint8)])).to(dev)
        d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
        d_counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
        g.search_hits(d_chars.data_ptr(), d_off.data_ptr(), 0, Q, 0, d_counts.data_ptr())
        torch.cuda.synchronize()
        assert g.last_ordered_kernel_is_lookup() == (name == "random" or not seed_order), name
        assert np.array_equal(d_counts.cpu().numpy().view(np.uint32), cnt), name
        if not seed_order and name == "planted":  # ... unless told otherwise: the general kernel, same counts
            monkeypatch.setenv("AWFM_GPU_MIXED_LOOKUP", "0")
            d_counts.fill_(7)
            g.search_hits(d_chars.data_ptr(), d_off.data_ptr(), 0, Q, 0, d_counts.data_ptr())
            torch.cuda.synchronize()
            assert np.array_equal(d_counts.cpu().numpy().view(np.uint32), cnt), "general kernel"
            monkeypatch.delenv("AWFM_GPU_MIXED_LOOKUP")
    g.destroy()
    ix.dealloc()


def test_mixed_length_lookup_prediction_and_whole_line_counts(oracle, awfm, require_gpu, wide, monkeypatch):
    """Round 5, mixed-length batches: the sample's verdict goes to page-locked host memory as for fixed lengths, and a batch
    whose predecessors agree launches one front end (awfmGpuLastLookupFront: 1 = mixedLookupSearchKernel alone, which then
    takes whatever the batch is; 2 = the 16-byte-record path alone).  With the lookup kernel alone a dense search is not
    pre-filled: the kernel stores every k-mer's count (and range) itself, a round's at a time in whole lines, survivors' final
    ranges out of their slots.  Counts and list against the oracle
    whatever was predicted, the batch the prediction is wrong for included."""
    import torch
    for name in ("AWFM_GPU_MIXED_LOOKUP", "AWFM_GPU_LOOKUP_PREDICT"):
        monkeypatch.delenv(name, raising=False)
    n, Q = 300000, (1 << 20) + 77
    txt = synth.text(n + 47, n, synth.DNA_ALPHABET).copy()
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    g.set_deep_seed(12)
    dev = torch.device("cuda")
    rng = np.random.default_rng(6)
    lengths = rng.integers(14, 33, Q)  # (longer than the deeper table: drawn from the text they survive their entry)
    lengths[::7] = rng.integers(0, 36, len(lengths[::7]))  # ... and none, 1..32, too long for the tables
    offsets = np.zeros(Q + 1, np.uint64)
    np.cumsum(lengths, out=offsets[1:])
    total = int(offsets[-1])
    starts = rng.integers(0, n - 40, Q)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    batches = {}
    for name in ("random", "planted"):
        if name == "random":
            chars = np.frombuffer(synth.DNA_ALPHABET, np.uint8)[rng.integers(0, 4, total)].copy()
        else:
            idx = np.repeat(starts, lengths) + (np.arange(total) - np.repeat(offsets[:-1].astype(np.int64), lengths))
            chars = txt[idx].copy()
        chars[::5000] = ord("n")  # some for the general kernel
        sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=4)
        batches[name] = (torch.from_numpy(np.concatenate([chars, np.zeros(8, np.uint8)])).to(dev), sp, ep, cnt)
    d_counts = torch.empty(Q, dtype=torch.int32, device=dev)
    d_kmers = torch.zeros(Q, dtype=torch.int32, device=dev)
    d_ranges = torch.zeros(Q * 2, dtype=torch.int64, device=dev)
    d_num = torch.zeros(1, dtype=torch.int32, device=dev)

    def run(name, listed):
        d_chars, sp, ep, cnt = batches[name]
        if listed:
            g.search_hits_compact(d_chars.data_ptr(), d_off.data_ptr(), 0, Q, d_kmers.data_ptr(), d_ranges.data_ptr(), Q, d_num.data_ptr())
            torch.cuda.synchronize()
            m = int(d_num.item())
            has = np.flatnonzero(cnt > 0)
            assert m == len(has), (name, m, len(has))
            ids = d_kmers[:m].cpu().numpy().view(np.uint32)
            order = np.argsort(ids)
            r = d_ranges[: 2 * m].cpu().numpy().view(np.uint64).reshape(m, 2)[order]
            assert np.array_equal(ids[order], has) and np.array_equal(r[:, 0], sp[has]) and np.array_equal(r[:, 1], ep[has]), name
        elif listed == "ranges":  # dense, with ranges (the hits-only contract: the exact range of a k-mer with hits, an empty one otherwise)
            d_dense = torch.full((2 * Q,), 5, dtype=torch.int64, device=dev)
            d_counts.fill_(7)
            g.search_hits(d_chars.data_ptr(), d_off.data_ptr(), 0, Q, d_dense.data_ptr(), d_counts.data_ptr())
            torch.cuda.synchronize()
            r = d_dense.cpu().numpy().view(np.uint64).reshape(Q, 2)
            has = cnt > 0
            assert np.array_equal(r[has, 0], sp[has]) and np.array_equal(r[has, 1], ep[has]) and np.all(r[~has, 0] > r[~has, 1]), name
            assert np.array_equal(d_counts.cpu().numpy().view(np.uint32), cnt), name
        else:
            d_counts.fill_(7)
            g.search_hits(d_chars.data_ptr(), d_off.data_ptr(), 0, Q, 0, d_counts.data_ptr())
            torch.cuda.synchronize()
            got = d_counts.cpu().numpy().view(np.uint32)
            assert np.array_equal(got, cnt), (name, int(np.flatnonzero(got != cnt)[0]))
        return g.last_lookup_front()

    fronts = [run("random", False), run("random", True), run("random", False), run("random", True)]
    assert fronts[0] == 0 and fronts[2:] == [1, 1], fronts
    assert run("random", "ranges") == 1  # (whole-line ranges)
    assert run("planted", "ranges") == 1  # predicted from the random batches: the lookup kernel takes the planted one, whole-line ranges
    held = [run("planted", ("ranges", True, False)[i % 3]) for i in range(9)]
    assert held[:8] == [0] * 8, held  # the miss switched the prediction off for eight searches
    later = [run("planted", i % 2 == 0) for i in range(3)]
    assert later[-1] == 2, (held, later)  # ... and then the 16-byte-record path alone
    assert not g.last_ordered_kernel_is_lookup()
    assert run("random", True) == 2  # (wrong the other way round: slower, the same list)
    assert run("random", False) == 0
    monkeypatch.setenv("AWFM_GPU_LOOKUP_PREDICT", "0")
    assert [run("random", False), run("random", True)] == [0, 0]
    g.destroy()
    ix.dealloc()


def test_ordered_search_when_every_wave_takes_many_chunks(oracle, awfm, require_gpu, wide):
    """batches large enough that every wave of the ordered kernel draws several tickets (its record prefetch runs
    ahead of the k-mer being searched): 600 000 mixed-length and 600 000 fixed-length k-mers, device generators,
    counts / hit ranges / hit offsets / positions against the oracle"""
    import torch
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda")
    n, Q = 3_000_000, 600_000
    d_text = torch.empty(n, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 9, 0, None) == 1
    ix = awfm.gpu_create_index(d_text.data_ptr(), awfm.AwFmAlphabetDna, 8, 9, on_device_length=n)
    oi = oracle.Index.wrap(oracle.DNA, 8, 9, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix, acquire=True)
    g.set_ordered(1)
    d_scratch = torch.zeros(awfm.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
    for mixed in (True, False):
        if mixed:
            d_len = torch.empty(Q, dtype=torch.int64, device=dev)
            assert L.awfmGpuSynthMixedLengths(d_len.data_ptr(), 0, Q, 5, 32, 205, None) == 1
            d_off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
            torch.cumsum(d_len, 0, out=d_off[1:])
            d_chars = torch.empty(int(d_off[-1]) + 8, dtype=torch.uint8, device=dev)
            assert L.awfmGpuSynthMixedQueries(d_chars.data_ptr(), d_off.data_ptr(), 0, Q, 205, d_text.data_ptr(), n, 0, None) == 1
            off_ptr, K = d_off.data_ptr(), 0
            offsets = d_off.cpu().numpy().view(np.uint64)
            chars = d_chars[: int(offsets[-1])].cpu().numpy()
        else:
            K = 19
            d_chars = torch.empty(Q * K + 8, dtype=torch.uint8, device=dev)
            assert L.awfmGpuSynthRandomQueries(d_chars.data_ptr(), 0, Q // 2, K, 206, 0, None) == 1
            assert L.awfmGpuSynthPlantedQueries(d_chars.data_ptr() + (Q // 2) * K, Q // 2, Q - Q // 2, K, 207,
                                                d_text.data_ptr(), n, None) == 1
            off_ptr = 0
            chars = d_chars[: Q * K].cpu().numpy()
            offsets = np.arange(Q + 1, dtype=np.uint64) * np.uint64(K)
        sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=8)
        hit_off, pos, _ = oi.batch_locate(sp, ep, threads=8)
        d_ranges = torch.full((Q * 2,), 7, dtype=torch.int64, device=dev)
        d_counts = torch.full((Q,), 7, dtype=torch.int32, device=dev)
        assert g.search_hits_is_ordered(mixed, K, Q)
        g.search_hits(d_chars.data_ptr(), off_ptr, K, Q, d_ranges.data_ptr(), d_counts.data_ptr())
        torch.cuda.synchronize()
        _check_hits_contract(d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2), d_counts.cpu().numpy().view(np.uint32),
                             sp, ep, cnt)
        d_hit_off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
        total = g.hit_offsets_from_counts(d_counts.data_ptr(), Q, d_hit_off.data_ptr(), d_scratch.data_ptr())
        assert total == len(pos) and np.array_equal(d_hit_off.cpu().numpy().view(np.uint64), hit_off)
        d_pos = torch.zeros(max(total, 1), dtype=torch.int64, device=dev)
        g.locate(d_ranges.data_ptr(), d_hit_off.data_ptr(), Q, total, d_pos.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d_pos[:total].cpu().numpy().view(np.uint64), pos)
    g.destroy()
    ix.dealloc()


def test_rna_alphabet_and_u_for_t(oracle, awfm, require_gpu, wide):
    """AwFmAlphabetRna indices (ref src/AwFmIndex.h:30-34, src/AwFmLetter.c:4-22: u and t are the same letter):
    text and queries written with u, t or a mix give the ranges and positions of the oracle's RNA index, through
    the general kernel and through the ordered path"""
    import torch
    txt = synth.text(41, 200000).copy()
    rna = txt.copy()
    rna[rna == ord("t")] = ord("u")
    rna[::7][rna[::7] == ord("u")] = ord("U")  # some upper case (a text that mixes t and u sorts them as two bytes
    # in the reference too, src/AwFmLetter.c:33-36: not a consistent index, not tested)
    ix = awfm.create_index(rna, awfm.AwFmAlphabetRna, 8, 7)
    oi = oracle.Index.wrap(oracle.RNA, 8, 7, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    ref = oracle.Index.from_text(txt.tobytes(), oracle.DNA, 8, 7)  # the same text as DNA: identical arrays
    assert np.array_equal(ix.blocks(), ref.blocks()) and np.array_equal(ix.seed_table(), ref.seed_table())
    g = awfm.GpuIndex(ix)
    Q, K = 20000, 15
    q = np.concatenate([synth.random_queries(42, Q // 2, K), synth.planted_queries(43, Q - Q // 2, K, txt)]).copy()
    flat = q.reshape(-1)
    rng = np.random.default_rng(44)
    sel = (flat == ord("t")) & (rng.random(flat.size) < 0.6)
    flat[sel] = ord("u")
    sel = (flat == ord("u")) & (rng.random(flat.size) < 0.3)
    flat[sel] = ord("U")
    chars, offsets = synth.fixed_csr(q)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    assert cnt.sum() >= Q // 2
    ranges, hoff, positions = g.locate_host(chars, offsets)
    assert np.array_equal(ranges[:, 0], sp) and np.array_equal(ranges[:, 1], ep)
    assert np.array_equal(hoff, hit_off) and np.array_equal(positions, pos)
    g.set_ordered(1)
    dev = torch.device("cuda")
    d_chars = torch.from_numpy(chars).to(dev)
    d_ranges = torch.zeros(Q * 2, dtype=torch.int64, device=dev)
    d_counts = torch.zeros(Q, dtype=torch.int32, device=dev)
    g.search_hits(d_chars.data_ptr(), 0, K, Q, d_ranges.data_ptr(), d_counts.data_ptr())
    torch.cuda.synchronize()
    _check_hits_contract(d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2), d_counts.cpu().numpy().view(np.uint32),
                         sp, ep, cnt)
    g.destroy()
    ix.dealloc()


@pytest.mark.timeout(300)
def test_deep_seed_env_knob_through_the_drop_in_api(oracle, awfm, require_gpu, monkeypatch, tmp_path):
    """$AWFM_GPU_DEEP_SEED_K is read when a device image is created.  awFmParallelSearchLocate creates that image
    lazily while it holds the image-table lock (an index loaded with awFmReadIndexFromFile has none yet): this used
    to self-deadlock.  The knob also applies to the image a GPU-built index adopts, and a device list that changes
    between calls gets that device's own image (entries are keyed by device, not by list position)."""
    import ctypes as C
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    txt = synth.text(91, 150000)
    chars, offsets = synth.mixed_queries(92, 4001, txt, synth.DNA_ALPHABET, 5, 30)
    kmers = [chars[int(offsets[i]):int(offsets[i + 1])].tobytes() for i in range(4001)]
    oi = oracle.Index.from_text(txt.tobytes(), oracle.DNA, 8, 6)
    sp, ep, cnt, _ = oi.batch_search(chars, offsets)
    hit_off, pos, _ = oi.batch_locate(sp, ep)
    path = str(tmp_path / "deep.awfmi")
    awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 6, file_src=path).dealloc()
    monkeypatch.setenv("AWFM_GPU_DEEP_SEED_K", "9")

    def check(ix):
        lst = awfm.KmerSearchList(4001)
        lst.fill(kmers)
        assert awfm.parallel_search_locate(ix, lst, 4) == awfm.AwFmSuccess
        assert L.awfmGpuLastBatchStatus() == awfm.AwFmSuccess
        assert np.array_equal(lst.counts(), cnt)
        for i in range(0, 4001, 13):
            assert np.array_equal(lst.positions(i), pos[int(hit_off[i]):int(hit_off[i + 1])])
        lst.dealloc()
        img = L.awfmGpuIndexAcquire(ix.ptr)
        plain = awfm.GpuIndex(ix)  # a second image of the same index, created without going through the table
        deep_bytes = 4 ** 9 * 8  # {sp, length} entries below 2^32 positions
        assert L.awfmGpuIndexDeviceBytes(C.c_void_p(img)) >= deep_bytes  # the table was really built
        plain.destroy()

    ix = awfm.read_index_from_file(path)
    check(ix)
    # the same index, now asked for with an explicit device list: device 0's image (and a lane on it)
    monkeypatch.setenv("AWFM_GPU_DEVICES", "0,0")
    check(ix)
    imgs = (C.c_void_p * 4)()
    assert L.awfmGpuIndexAcquireAll(ix.ptr, imgs, 4) == 2 and imgs[0] != imgs[1]
    assert L.awfmGpuIndexDevice(C.c_void_p(imgs[0])) == 0 and L.awfmGpuIndexDevice(C.c_void_p(imgs[1])) == 0
    ix.dealloc()
    monkeypatch.delenv("AWFM_GPU_DEVICES")
    ix = awfm.gpu_create_index(txt, awfm.AwFmAlphabetDna, 8, 6)  # the builder's adopted image gets the table too
    check(ix)
    ix.dealloc()
