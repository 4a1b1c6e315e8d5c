"""CPU-side tests of libawfmindex_amd.so: exported symbols, struct ABI, host builder, .awfmi file,
single-query API, search-list ownership -- no GPU compute.

Reference behaviour mirrored: test/createTests/AwFmCreationTest.c, test/fileTests/AwFmFileTests.c:32-381,
test/searchTest/searchTest.c:124-200, test/staticLibTest, src/AwFmParallelSearch.c:36-93.
"""
import ctypes as C
import os

import numpy as np
import pytest

from avxwindowfmindex_amd import synth


def test_library_exports_every_declared_symbol(awfm):
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    missing = [s for s in _lib.API_SYMBOLS + _lib.GPU_SYMBOLS if not hasattr(L, s)]
    assert not missing
    # every function declared in the two public headers is in the lists the loader checks
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    declared = set()
    for h in ("include/AwFmIndex.h", "include/awfm_gpu.h"):
        src = open(os.path.join(root, h)).read()
        declared |= set(re.findall(r"\b(awFm[A-Z]\w+|awfmGpu\w+|awfmPack\w+)\s*\(", src))
    assert declared <= set(_lib.API_SYMBOLS + _lib.GPU_SYMBOLS), declared - set(_lib.API_SYMBOLS + _lib.GPU_SYMBOLS)


def test_struct_abi_matches_reference_layout(awfm):
    """LP64 sizes/offsets of the reference structs (probed from the compiled reference)"""
    from avxwindowfmindex_amd import _lib
    assert C.sizeof(_lib.AwFmKmerSearchData) == 32
    assert _lib.AwFmKmerSearchData.positionList.offset == 16 and _lib.AwFmKmerSearchData.count.offset == 24
    assert _lib.AwFmKmerSearchData.capacity.offset == 28
    assert C.sizeof(_lib.AwFmKmerSearchList) == 24 and C.sizeof(_lib.AwFmSearchRange) == 16
    assert C.sizeof(_lib.AwFmIndexConfiguration) == 12
    assert _lib.AwFmIndexConfiguration.alphabetType.offset == 4
    assert _lib.AwFmIndexConfiguration.keepSuffixArrayInMemory.offset == 8
    assert C.sizeof(_lib.AwFmIndex) == 112


@pytest.mark.parametrize("alphabet_name", ["dna", "amino"])
def test_host_builder_is_byte_identical_to_the_oracle(oracle, awfm, alphabet_name):
    amino = alphabet_name == "amino"
    alpha, oalpha = (awfm.AwFmAlphabetAmino, oracle.AMINO) if amino else (awfm.AwFmAlphabetDna, oracle.DNA)
    letters = synth.AMINO_ALPHABET if amino else synth.DNA_ALPHABET
    for n in (0, 1, 29, 255, 256, 257, 5000, 70000):
        for ratio, k in ((1, 1), (3, 2), (8, 2 if amino else 6)):
            raw = synth.text(n + 1, n, letters).copy()
            if n > 100:
                raw[5:9] = ord("x")
                raw[50] = ord("N")
            ix = awfm.create_index(raw, alpha, ratio, k)
            oi = oracle.Index.from_text(raw.tobytes(), oalpha, ratio, k)
            assert np.array_equal(ix.blocks(), oi.blocks())
            assert np.array_equal(ix.prefix_sums(), oi.prefix_sums())
            assert np.array_equal(ix.seed_table(), oi.seed_table())
            assert np.array_equal(ix.packed_sa(), oi.packed_sa())
            ix.dealloc()


def test_host_builder_on_repetitive_texts(oracle, awfm):
    for raw in (b"a" * 3000, b"acgt" * 2000 + b"a", b"acgtacgtaa" * 1500 + b"t" * 100, b"$acg$t"):
        ix = awfm.create_index(np.frombuffer(raw, np.uint8), awfm.AwFmAlphabetDna, 4, 3)
        oi = oracle.Index.from_text(raw, oracle.DNA, 4, 3)
        assert np.array_equal(ix.blocks(), oi.blocks()) and np.array_equal(ix.packed_sa(), oi.packed_sa())
        assert np.array_equal(ix.seed_table(), oi.seed_table())
        ix.dealloc()


def test_awfmi_file_round_trip_and_layout(awfm, tmp_path):
    """ref src/AwFmFile.c:20-193: magic, header fields, section sizes; read-back equals what was written"""
    raw = synth.text(3, 29)
    for store in (False, True):
        path = str(tmp_path / f"toy_{int(store)}.awfmi")
        ix = awfm.create_index(raw, awfm.AwFmAlphabetDna, 8, 3, store_sequence=store, file_src=path)
        blob = open(path, "rb").read()
        # 10 magic + 12 header + 8 bwtLength + 1 block*160 + 6*8 prefix sums + 64*16 seed table (+29 sequence)
        # + packed SA: ceil(4 samples * 5 bits / 8) + 8 pad (1275 bytes for a 29-bp text)
        assert len(blob) == 10 + 12 + 8 + 160 + 48 + 1024 + (29 if store else 0) + 3 + 8
        assert blob[:10] == b"AwFmIndex\n"
        assert int.from_bytes(blob[10:14], "little") == 8 and blob[18] == 8 and blob[19] == 3 and blob[20] == 2
        assert blob[21] == int(store) and int.from_bytes(blob[22:30], "little") == 30
        if store:
            assert blob[30 + 160 + 48 + 1024:][:29] == raw.tobytes()
        for keep in (True, False):
            back = awfm.read_index_from_file(path, keep_sa_in_memory=keep)
            assert back.bwt_length == 30
            assert np.array_equal(back.blocks(), ix.blocks()) and np.array_equal(back.prefix_sums(), ix.prefix_sums())
            assert np.array_equal(back.seed_table(), ix.seed_table())
            if keep:
                assert np.array_equal(back.packed_sa(), ix.packed_sa())
            else:
                assert back.packed_sa() is None
            if store:
                from avxwindowfmindex_amd import _lib
                buf = C.create_string_buffer(11)
                assert _lib.lib().awFmReadSequenceFromFile(back.ptr, 5, 10, buf) == awfm.AwFmFileReadOkay
                assert buf.value == raw.tobytes()[5:15]
            back.dealloc()
        ix.dealloc()
    from avxwindowfmindex_amd import _lib
    out = C.POINTER(_lib.AwFmIndex)()
    assert _lib.lib().awFmReadIndexFromFile(C.byref(out), str(tmp_path / "missing.awfmi").encode(), True) == -10
    bad = tmp_path / "bad.awfmi"
    bad.write_bytes(b"NotAnIndex" + bytes(100))
    assert _lib.lib().awFmReadIndexFromFile(C.byref(out), str(bad).encode(), True) == -9


@pytest.mark.parametrize("alphabet_name", ["dna", "amino"])
def test_single_query_api_matches_oracle(oracle, awfm, alphabet_name, tmp_path):
    """awFmFindSearchRangeForString, step, backtrace, hit positions (in memory and from the file)"""
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    amino = alphabet_name == "amino"
    alpha, oalpha = (awfm.AwFmAlphabetAmino, oracle.AMINO) if amino else (awfm.AwFmAlphabetDna, oracle.DNA)
    letters = synth.AMINO_ALPHABET if amino else synth.DNA_ALPHABET
    raw = synth.text(17, 6000, letters).copy()
    raw[40:43] = ord("x")
    oi = oracle.Index.from_text(raw.tobytes(), oalpha, 7, 2)
    for keep in (True, False):
        ix = awfm.create_index(raw, alpha, 7, 2, keep_sa_in_memory=keep, file_src=str(tmp_path / f"s{int(keep)}.awfmi"))
        chars, offsets = synth.mixed_queries(18, 400, raw, letters, 1, 10)
        for j in range(400):
            kmer = bytes(chars[int(offsets[j]):int(offsets[j + 1])])
            assert ix.find_search_range_for_string(kmer) == oi.range_for_string(kmer)
        rng = np.random.default_rng(4)
        for _ in range(200):
            p = int(rng.integers(0, 6001))
            pos = C.c_uint64(p)
            f = L.awFmAminoBacktraceReturnPreviousLetterIndex if amino else L.awFmNucleotideBacktraceReturnPreviousLetterIndex
            letter = f(ix.ptr, C.byref(pos))
            if oi.letter_at(p) == (21 if amino else 5):
                assert letter == 0 and pos.value == p
            else:
                assert letter == oi.letter_at(p) and pos.value == oi.lf(p)
            rc = C.c_int(0)
            assert L.awFmFindDatabaseHitPositionSingle(ix.ptr, p, C.byref(rc)) == oi.locate_one(p)
            assert rc.value == awfm.AwFmFileReadOkay
        r = _lib.AwFmSearchRange(*oi.range_for_string(raw[100:103].tobytes()))
        rc = C.c_int(0)
        ptr = L.awFmFindDatabaseHitPositions(ix.ptr, C.byref(r), C.byref(rc))
        n = L.awFmSearchRangeLength(C.byref(r))
        got = [ptr[i] for i in range(n)]
        L.free(C.cast(ptr, C.c_void_p))
        assert got == [oi.locate_one(p) for p in range(r.startPtr, r.endPtr + 1)] and rc.value == awfm.AwFmFileReadOkay
        empty = _lib.AwFmSearchRange(5, 4)
        assert not L.awFmFindDatabaseHitPositions(ix.ptr, C.byref(empty), C.byref(rc)) and rc.value == -1
        step = L.awFmAminoIterativeStepBackwardSearch if amino else L.awFmNucleotideIterativeStepBackwardSearch
        rr = L.awFmCreateInitialQueryRangeFromChar(ix.ptr, bytes([letters[2]]))
        step(ix.ptr, C.byref(rr), 1)
        assert (rr.startPtr, rr.endPtr) == oi.step(int(oi.prefix_sums()[2]), int(oi.prefix_sums()[3]) - 1, 1)
        ix.dealloc()


def test_search_list_ownership_and_return_codes(awfm):
    """ref src/AwFmParallelSearch.c:36-93: 4-slot malloc'ed position lists, strings never owned"""
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    lst = awfm.KmerSearchList(100)
    c = lst.ptr.contents
    assert c.capacity == 100 and c.count == 0
    assert all(c.kmerSearchData[i].capacity == 4 and c.kmerSearchData[i].count == 0 and
               bool(c.kmerSearchData[i].positionList) and not c.kmerSearchData[i].kmerString for i in range(100))
    lst.fill([b"acgt", b"gg"])
    assert lst.ptr.contents.count == 2 and lst.ptr.contents.kmerSearchData[1].kmerLength == 2
    lst.dealloc()
    assert L.awFmReturnCodeIsFailure(-11) and not L.awFmReturnCodeIsFailure(3)
    assert L.awFmReturnCodeIsSuccess(1) and not L.awFmReturnCodeIsSuccess(-1)
    r = _lib.AwFmSearchRange(7, 9)
    assert L.awFmSearchRangeLength(C.byref(r)) == 3
    r = _lib.AwFmSearchRange(10, 9)
    assert L.awFmSearchRangeLength(C.byref(r)) == 0
    out = C.POINTER(_lib.AwFmIndex)()
    cfg = _lib.AwFmIndexConfiguration(8, 4, 2, True, False)
    assert L.awFmCreateIndex(C.byref(out), None, None, 0, b"x") == -4
    assert L.awFmCreateIndexFromFasta(C.byref(out), C.byref(cfg), b"/nonexistent/a.fa", b"a.awfmi") == -10  # AwFmFileOpenFail


def test_batch_search_fails_loudly_without_a_gpu(awfm):
    """the hot path has no CPU fallback: without a device Locate returns a failure code, Count leaves its code in
    awfmGpuLastBatchStatus(), and neither leaves a stale count behind (a failed search reports no hits)"""
    from avxwindowfmindex_amd import _lib
    if _lib.lib().awfmGpuDeviceCount() > 0:
        pytest.skip("a GPU is visible here")
    raw = synth.text(5, 2000)
    ix = awfm.create_index(raw, awfm.AwFmAlphabetDna, 8, 4)
    lst = awfm.KmerSearchList(8)
    lst.fill([raw[10:20].tobytes()])
    lst.ptr.contents.kmerSearchData[0].count = 77
    awfm.parallel_search_count(ix, lst, 2)
    assert lst.counts()[0] == 0
    assert _lib.lib().awfmGpuLastBatchStatus() == _lib.AwFmGeneralFailure
    lst.ptr.contents.kmerSearchData[0].count = 77
    assert awfm.parallel_search_locate(ix, lst, 2) == -1
    assert lst.counts()[0] == 0
    assert b"no HIP device" in _lib.lib().awfmGpuLastError()
    with pytest.raises(RuntimeError):
        awfm.GpuIndex(ix)
    lst.dealloc()
    ix.dealloc()


def test_gpu_entry_points_reject_null_arguments_without_touching_a_device(awfm):
    """the flat GPU entry points added for mixed-length batches answer a null image or null buffers with an error code and
    a message -- no device call, so this runs without a GPU; the reporting accessors of a null image say "nothing"."""
    import ctypes as C
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    out = (C.c_uint64 * 8)()
    assert L.awfmGpuMixedLookupLineTally(None, None, None, 0, C.byref(out)) == -4  # AwFmNullPtrError (include/AwFmIndex.h)
    assert b"awfmGpuMixedLookupLineTally" in L.awfmGpuLastError()
    assert L.awfmGpuIndexLengthTableBytes(None) == 0 and L.awfmGpuIndexLengthTableBuildSeconds(None) == 0.0
    assert L.awfmGpuSearchHits(None, None, None, 0, 5, None, None, None) == -4  # AwFmNullPtrError (include/AwFmIndex.h)


def test_reference_shared_library_program_relinks_unchanged(awfm, tmp_path):
    """oracle/_ref/sharedLibTest is the reference's own test/sharedLibTest/awfmiTest.c (it includes nothing but
    "AwFmIndex.h"), compiled where it lies against include/AwFmIndex.h and libawfmindex_amd.so by
    oracle/Makefile.  It builds an index from its fixture test.fa and must report success; the file it writes
    is then read back through this library."""
    import subprocess
    from avxwindowfmindex_amd import _lib
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "sharedLibTest")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/sharedLibTest is only built where /root/reference is mounted")
    # contents of the reference's fixture test/sharedLibTest/test.fa (data: one record)
    record = b"test1sequencedataagasfnlawebrfilqawhbefrilahwbseflikhabsdlfikhbas"
    (tmp_path / "test.fa").write_bytes(b">test 1 header\n" + record + b"\n")
    out = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "createIndex successful" in out.stdout, out.stdout + out.stderr
    ix = awfm.read_index_from_file(str(tmp_path / "output.awfmi"))
    L = _lib.lib()
    assert L.awFmGetNumSequences(ix.ptr) == 1 and ix.header(0) == b"test 1 header"
    assert ix.bwt_length == len(record) + 2  # the record, its terminator and the sentinel
    # the program's configuration is a DNA index (ratio 2, k 2): of the record's letters only a, c, g, t/u are
    # themselves, everything else is the ambiguity letter
    sp, ep = ix.find_search_range_for_string(b"ataaga")
    assert ep - sp + 1 == 1
    sp, ep = ix.find_search_range_for_string(b"gata")
    assert sp > ep
    ix.dealloc()


def test_pack_kmers_layout_and_rejections(awfm):
    """awfmPackKmers: first character most significant, 2 bits per nucleotide (the letter indices of ref
    src/AwFmLetter.c:4-22) / 5 bits per amino acid (ref src/AwFmLetter.c:55-67); what cannot be expressed is refused"""
    q = synth.random_queries(2, 2000, 21).copy()
    q[::3] &= 0xDF  # upper case packs like lower case
    ts = np.argwhere(q == ord("t"))[::2]
    q[ts[:, 0], ts[:, 1]] = ord("u")
    lut = np.zeros(256, np.uint64)
    for ch, v in ((b"c", 1), (b"g", 2), (b"t", 3), (b"u", 3)):
        lut[ch[0]] = lut[ch[0] & 0xDF] = v
    expect = np.zeros(len(q), np.uint64)
    for c in range(21):
        expect = (expect << np.uint64(2)) | lut[q[:, c]]
    assert np.array_equal(awfm.pack_kmers(q), expect)
    assert awfm.pack_kmers(np.frombuffer(b"t" * 32, np.uint8).reshape(1, 32))[0] == np.uint64(2**64 - 1)
    qa = synth.random_queries(3, 2000, 12, synth.AMINO_ALPHABET)
    index_of = np.zeros(256, np.uint64)
    for i, ch in enumerate(synth.AMINO_ALPHABET):
        index_of[ch] = i
    expect = np.zeros(len(qa), np.uint64)
    for c in range(12):
        expect = (expect << np.uint64(5)) | index_of[qa[:, c]]
    assert np.array_equal(awfm.pack_kmers(qa, awfm.AwFmAlphabetAmino), expect)
    for bad, alphabet in ((b"acgn", awfm.AwFmAlphabetDna), (b"ac$t", awfm.AwFmAlphabetDna), (b"acdx", awfm.AwFmAlphabetAmino),
                          (b"acdb", awfm.AwFmAlphabetAmino)):
        with pytest.raises(ValueError):
            awfm.pack_kmers(np.frombuffer(bad, np.uint8).reshape(1, 4), alphabet)
    with pytest.raises(ValueError):  # more characters than a word holds
        awfm.pack_kmers(np.full((1, 33), ord("a"), np.uint8))
    with pytest.raises(ValueError):
        awfm.pack_kmers(np.full((1, 13), ord("a"), np.uint8), awfm.AwFmAlphabetAmino)


def test_parallel_for_ranges_pool_and_concurrent_callers(awfm):
    """awfmParallelFor (awfm_threads.c): the ranges of a loop depend on (numThreads, n) alone -- the AoS packing pairs two
    loops by range number --, they tile [0, n), every range runs once, and the kept workers serve loops of different widths
    one after another; a second caller that finds the pool taken runs its loop on threads of its own."""
    import threading
    from avxwindowfmindex_amd import _lib
    L = _lib.lib()
    RANGE = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint)
    L.awfmParallelFor.argtypes = [C.c_uint, C.c_uint64, RANGE, C.c_void_p]
    L.awfmParallelFor.restype = None

    def run(threads, n):
        seen, lock = [], threading.Lock()

        def body(_ctx, begin, end, tid):
            with lock:
                seen.append((tid, begin, end))
        cb = RANGE(body)
        L.awfmParallelFor(threads, n, cb, None)
        return sorted(seen)

    for threads, n in ((8, 100000), (3, 4096), (8, 100001), (32, 70001), (5, 100000), (8, 100000), (64, 5000), (1, 9999)):
        got = run(threads, n)
        if threads == 1:
            assert got == [(0, 0, n)]
            continue
        chunk = (n + min(threads, 64) - 1) // min(threads, 64)
        want = [(t, t * chunk, min(n, (t + 1) * chunk)) for t in range(min(threads, 64)) if t * chunk < n]
        assert got == want, (threads, n)
    assert run(8, 100) == [(0, 0, 100)]  # short loops run on the caller
    # two callers at once: both loops complete with their own ranges
    results = {}

    def caller(name, threads, n):
        results[name] = run(threads, n)
    ts = [threading.Thread(target=caller, args=(i, 4 + i, 50000 + i)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(4):
        threads, n = 4 + i, 50000 + i
        chunk = (n + threads - 1) // threads
        assert results[i] == [(t, t * chunk, min(n, (t + 1) * chunk)) for t in range(threads) if t * chunk < n]
    # a forked child starts over with an empty pool
    pid = os.fork()
    if pid == 0:
        ok = run(6, 60000) == [(t, t * 10000, (t + 1) * 10000) for t in range(6)]
        os._exit(0 if ok else 1)
    assert os.waitpid(pid, 0)[1] == 0
