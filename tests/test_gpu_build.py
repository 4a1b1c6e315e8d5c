"""GPU index construction and device-side synthetic generators against their host twins.

The arrays an index is made of (blocks, prefix sums, seed table, packed sampled SA) must be byte
identical whichever builder made them (ref test/createTests/AwFmCreationTest.c:151-295,
test/bwtTest/bwtTest.c:95-213, test/kmerSeedTableTests/kmerSeedTableTests.c:203-228 pin them against
brute force; the host builder is pinned to the oracle's in tests/test_host_lib.py).
"""
import numpy as np
import pytest

from avxwindowfmindex_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[False, True], ids=["pos32", "pos64"])
def build_wide(request, diag):
    """the suffix sort of the GPU builder with 32-bit and with 64-bit positions and ranks (the latter is what texts of
    2^32 - 1 characters and more get; $AWFM_GPU_DIAG build_wide=1 selects it on any text)"""
    diag(build_wide="1" if request.param else None)
    return request.param


def _host_build(awfm, *args, **kwargs):
    """awFmCreateIndex kept on the host builder (texts of 2^20 characters and more go to the GPU builder otherwise)"""
    import os
    os.environ["AWFM_HOST_BUILD"] = "1"
    try:
        return awfm.create_index(*args, **kwargs)
    finally:
        del os.environ["AWFM_HOST_BUILD"]


def _same_arrays(a, b):
    assert a.bwt_length == b.bwt_length
    assert np.array_equal(a.prefix_sums(), b.prefix_sums()), "prefix sums"
    assert np.array_equal(a.blocks(), b.blocks()), "BWT blocks"
    assert np.array_equal(a.seed_table(), b.seed_table()), "seed table"
    assert np.array_equal(a.packed_sa(), b.packed_sa()), "packed sampled SA"


@pytest.mark.parametrize("n", [0, 1, 2, 29, 255, 256, 257, 4096, 100003, 1 << 20])
def test_dna_build_matches_host(awfm, require_gpu, build_wide, n):
    txt = synth.text(40 + n, n).copy()
    if n > 100:
        txt[7:12] = ord("N")
        txt[n // 2] = ord("x")
    for ratio, k in ((1, 1), (8, 6), (3, 3)):
        host = _host_build(awfm, txt, awfm.AwFmAlphabetDna, ratio, k)
        dev = awfm.gpu_create_index(txt, awfm.AwFmAlphabetDna, ratio, k)
        _same_arrays(host, dev)
        host.dealloc()
        dev.dealloc()


@pytest.mark.parametrize("n", [0, 1, 31, 256, 5000, 300001])
def test_amino_build_matches_host(awfm, require_gpu, build_wide, n):
    txt = synth.text(50 + n, n, synth.AMINO_ALPHABET).copy()
    if n > 100:
        txt[3:6] = ord("x")
        txt[n // 2] = ord("b")
    for ratio, k in ((1, 1), (8, 3), (5, 2)):
        host = _host_build(awfm, txt, awfm.AwFmAlphabetAmino, ratio, k)
        dev = awfm.gpu_create_index(txt, awfm.AwFmAlphabetAmino, ratio, k)
        _same_arrays(host, dev)
        host.dealloc()
        dev.dealloc()


def test_repetitive_texts_need_doubling_rounds(awfm, require_gpu, build_wide):
    cases = [b"a" * 5000, b"acgt" * 4000 + b"a", (b"acgtacgtaa" * 3000) + b"t" * 300,
             bytes(synth.text(3, 2000)) * 40]
    for raw in cases:
        txt = np.frombuffer(raw, np.uint8)
        host = _host_build(awfm, txt, awfm.AwFmAlphabetDna, 4, 4)
        dev = awfm.gpu_create_index(txt, awfm.AwFmAlphabetDna, 4, 4)
        _same_arrays(host, dev)
        host.dealloc()
        dev.dealloc()


def test_gpu_built_index_searches_with_its_resident_image(oracle, awfm, require_gpu, tmp_path):
    txt = synth.text(61, 400000)
    path = str(tmp_path / "gpu_built.awfmi")
    ix = awfm.gpu_create_index(txt, awfm.AwFmAlphabetDna, 8, 8, file_src=path)
    g = awfm.GpuIndex(ix, acquire=True)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(),
                           ix.packed_sa())
    q = np.concatenate([synth.random_queries(62, 3000, 15), synth.planted_queries(63, 3000, 15, txt)])
    chars, offsets = synth.fixed_csr(q)
    sp, ep, _, _ = oi.batch_search(chars, offsets)
    ho, pos, _ = oi.batch_locate(sp, ep)
    r, ho2, pos2 = g.locate_host(chars, offsets)
    assert np.array_equal(r[:, 0], sp) and np.array_equal(r[:, 1], ep)
    assert np.array_equal(ho2, ho) and np.array_equal(pos2, pos)
    # the file the GPU builder wrote loads back to the same arrays
    back = awfm.read_index_from_file(path)
    _same_arrays(ix, back)
    back.dealloc()
    ix.dealloc()


def test_device_generators_match_numpy(awfm, require_gpu):
    import torch
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    n = 100000
    for amino, alphabet in ((0, synth.DNA_ALPHABET), (1, synth.AMINO_ALPHABET)):
        d = torch.empty(n, dtype=torch.uint8, device="cuda")
        assert L.awfmGpuSynthText(d.data_ptr(), 0, n, 11, amino, None) == 1
        torch.cuda.synchronize()
        txt = synth.text(11, n, alphabet)
        assert np.array_equal(d.cpu().numpy(), txt)
        part = torch.empty(1000, dtype=torch.uint8, device="cuda")
        assert L.awfmGpuSynthText(part.data_ptr(), 5000, 1000, 11, amino, None) == 1
        assert np.array_equal(part.cpu().numpy(), txt[5000:6000])
        q = torch.empty(2000 * 21, dtype=torch.uint8, device="cuda")
        assert L.awfmGpuSynthRandomQueries(q.data_ptr(), 100, 2000, 21, 12, amino, None) == 1
        assert np.array_equal(q.cpu().numpy().reshape(2000, 21), synth.random_queries(12, 2000, 21, alphabet, first=100))
        assert L.awfmGpuSynthPlantedQueries(q.data_ptr(), 100, 2000, 21, 13, d.data_ptr(), n, None) == 1
        assert np.array_equal(q.cpu().numpy().reshape(2000, 21), synth.planted_queries(13, 2000, 21, txt, first=100))


def test_device_mixed_generator_matches_numpy(awfm, require_gpu):
    import torch
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    n, count = 50000, 3000
    txt = synth.text(15, n)
    d_text = torch.from_numpy(txt.copy()).cuda()
    chars, offsets = synth.mixed_queries(16, count, txt, synth.DNA_ALPHABET, 8, 30, first=40)
    d_len = torch.empty(count, dtype=torch.int64, device="cuda")
    assert L.awfmGpuSynthMixedLengths(d_len.data_ptr(), 40, count, 8, 30, 16, None) == 1
    d_off = torch.zeros(count + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(d_len, 0, out=d_off[1:])
    assert np.array_equal(d_off.cpu().numpy().view(np.uint64), offsets)
    d_chars = torch.empty(int(d_off[-1].item()), dtype=torch.uint8, device="cuda")
    assert L.awfmGpuSynthMixedQueries(d_chars.data_ptr(), d_off.data_ptr(), 40, count, 16, d_text.data_ptr(), n, 0, None) == 1
    assert np.array_equal(d_chars.cpu().numpy(), chars)


def test_drop_in_create_index_goes_to_the_gpu_builder_and_writes_the_same_file(awfm, require_gpu, tmp_path, monkeypatch):
    """awFmCreateIndex / awFmCreateIndexFromFasta of a text of 2^20 characters and more are built on the GPU: the
    arrays AND the .awfmi file are byte-identical to the host builder's ($AWFM_HOST_BUILD=1), with the sequence
    stored or not, the SA kept in memory or not, from a plain text and from a multi-record FASTA file"""
    from avxwindowfmindex_amd import _lib
    n = (1 << 20) + 12345
    txt = synth.text(71, n).copy()
    txt[1000:1100] = ord("N")
    for keep_sa, store_seq in ((True, False), (False, True)):
        files = {}
        for where in ("gpu", "host"):
            if where == "host":
                monkeypatch.setenv("AWFM_HOST_BUILD", "1")
            else:
                monkeypatch.delenv("AWFM_HOST_BUILD", raising=False)
            f = str(tmp_path / f"{where}_{int(keep_sa)}{int(store_seq)}.awfmi")
            ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 7, keep_sa_in_memory=keep_sa, store_sequence=store_seq,
                                   file_src=f)
            # the GPU build leaves its device image registered for the batch search; the host build has none yet
            files[where] = (open(f, "rb").read(), ix.blocks().copy(), ix.seed_table().copy(), ix.prefix_sums().copy())
            ix.dealloc()
        assert files["gpu"][0] == files["host"][0], "index files differ"
        for a, b in zip(files["gpu"][1:], files["host"][1:]):
            assert np.array_equal(a, b)
    # FASTA: three records, wrapped lines
    recs = [synth.text(80 + i, 400000 + 1000 * i).tobytes() for i in range(3)]
    fa = tmp_path / "big.fa"
    with open(fa, "wb") as out:
        for i, r in enumerate(recs):
            out.write(b">record %d some description\n" % i)
            for at in range(0, len(r), 70):
                out.write(r[at:at + 70] + b"\n")
    blobs = {}
    for where in ("gpu", "host"):
        if where == "host":
            monkeypatch.setenv("AWFM_HOST_BUILD", "1")
        else:
            monkeypatch.delenv("AWFM_HOST_BUILD", raising=False)
        f = str(tmp_path / f"fa_{where}.awfmi")
        ix = awfm.create_index_from_fasta(str(fa), awfm.AwFmAlphabetDna, 8, 6, file_src=f)
        assert _lib.lib().awFmGetNumSequences(ix.ptr) == 3 and ix.header(2) == b"record 2 some description"
        assert ix.local_position(len(recs[0]) + 1 + 5) == (1, 5)
        blobs[where] = open(f, "rb").read()
        ix.dealloc()
    assert blobs["gpu"] == blobs["host"], "FASTA index files differ"


def test_genome_shaped_text_generator_and_build(awfm, require_gpu):
    """the genome-shaped synthetic text (repeat families, tandem repeats, runs of N: what GRCh38 is and a uniform text is
    not): the device generator writes the characters synth.genome_text defines, and the GPU builder's arrays are byte
    identical to the host builder's on 16 Mbp of it -- N runs of up to 2.6*10^5 characters (suffixes that tie over their
    whole length), 10^3 copies of a 300-character family, 6000-character units, tandem repeats"""
    import torch
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    for n in (5000, 300_001):
        d = torch.empty(n, dtype=torch.uint8, device="cuda")
        assert L.awfmGpuSynthGenomeText(d.data_ptr(), n, 7, None) == 1
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy(), synth.genome_text(7, n))
    n = 1 << 24
    txt = synth.genome_text(2, n)
    counts = {chr(c): int(k) for c, k in zip(*np.unique(txt, return_counts=True))}
    assert set(counts) == set("acgtn") and 0.02 < counts["n"] / n < 0.4
    d = torch.from_numpy(txt).cuda()
    q = torch.empty(3000 * 21, dtype=torch.uint8, device="cuda")
    assert L.awfmGpuSynthPlantedQueriesClean(q.data_ptr(), 5, 3000, 21, 13, d.data_ptr(), n, None) == 1
    assert np.array_equal(q.cpu().numpy().reshape(3000, 21), synth.planted_queries_clean(13, 3000, 21, txt, first=5))
    # k-mers drawn from the unique sequence only (no repeat family's block, no N): the offsets synth.py defines, letters only
    offs = torch.empty(3000, dtype=torch.int64, device="cuda")
    assert L.awfmGpuSynthPlantedQueriesUnique(q.data_ptr(), 5, 3000, 21, 13, d.data_ptr(), n, 2, offs.data_ptr(), None) == 1
    want = synth.planted_unique_offsets(13, 3000, 21, txt, 2, first=5)
    assert np.array_equal(offs.cpu().numpy().view(np.uint64), want)
    got = q.cpu().numpy().reshape(3000, 21)
    assert np.array_equal(got, txt[want.astype(np.int64)[:, None] + np.arange(21)[None, :]]) and np.all(np.isin(got, np.frombuffer(b"acgt", np.uint8)))
    host = _host_build(awfm, txt, awfm.AwFmAlphabetDna, 8, 8)
    dev = awfm.gpu_create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    _same_arrays(host, dev)
    host.dealloc()
    dev.dealloc()
