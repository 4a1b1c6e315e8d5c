"""The reference's threading contract: the index is read-only during a search, every query's slot is written
by exactly one thread, and callers may run concurrent searches on ONE index with DIFFERENT lists
(ref src/AwFmParallelSearch.c:103-129: every 8-query block writes its own slots; :167 the same for counting).

Here the searches of concurrent callers meet on one device image: its lanes, its scratch slots, its sort temporaries.  These
tests start several host threads on one index and require every caller's results to be the oracle's."""
import threading

import numpy as np
import pytest

from avxwindowfmindex_amd import synth

pytestmark = pytest.mark.gpu


def test_four_host_threads_search_one_index_with_their_own_lists(oracle, awfm, require_gpu):
    """awFmParallelSearchLocate / awFmParallelSearchCount from four threads at once, one index, four lists (two located, two
    counted; uniform and mixed lengths), several rounds: counts and every position list against the oracle"""
    txt = synth.text(401, 400000)
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    n = 70003  # above the size from which a list is dealt to the image's lanes in chunks
    jobs = []
    for t in range(4):
        if t % 2 == 0:
            q = np.concatenate([synth.random_queries(410 + t, n // 2, 14), synth.planted_queries(420 + t, n - n // 2, 14, txt)])
            kmers = [bytes(r) for r in q]
        else:
            chars, offsets = synth.mixed_queries(430 + t, n, txt, synth.DNA_ALPHABET, 6, 30)
            kmers = [chars[int(offsets[i]):int(offsets[i + 1])].tobytes() for i in range(n)]
        sp, ep, cnt, _ = oi.search_list(kmers)
        hit_off, pos, _ = oi.batch_locate(sp, ep)
        lst = awfm.KmerSearchList(n)
        lst.fill(kmers)
        jobs.append({"kmers": kmers, "cnt": cnt, "hit_off": hit_off, "pos": pos, "list": lst, "locate": t < 2, "errors": []})

    def work(job):
        try:
            for _ in range(3):
                if job["locate"]:
                    rc = awfm.parallel_search_locate(ix, job["list"], 4)
                    if rc != awfm.AwFmSuccess:
                        raise AssertionError(f"awFmParallelSearchLocate returned {rc}")
                else:
                    awfm.parallel_search_count(ix, job["list"], 4)
                if not np.array_equal(job["list"].counts(), job["cnt"]):
                    raise AssertionError("counts differ from the oracle")
                if job["locate"]:
                    for i in range(0, n, 41):
                        if not np.array_equal(job["list"].positions(i), job["pos"][int(job["hit_off"][i]):int(job["hit_off"][i + 1])]):
                            raise AssertionError(f"positions of k-mer {i} differ from the oracle")
        except Exception as e:  # noqa: BLE001  (reported by the main thread)
            job["errors"].append(repr(e))

    threads = [threading.Thread(target=work, args=(job,)) for job in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not any(t.is_alive() for t in threads), "a caller did not come back"
    for i, job in enumerate(jobs):
        assert not job["errors"], f"caller {i}: {job['errors']}"
        job["list"].dealloc()
    ix.dealloc()


@pytest.mark.parametrize("callers", [2, 3])
def test_callers_on_their_own_streams_share_one_image(oracle, awfm, require_gpu, callers):
    """the device-buffer API from several threads, each on its own stream with its own buffers, the seed-order path forced
    on: dense results (awfmGpuSearchHits) and the listed pipeline (SearchHitsCompact -> SortHitsOnDevice ->
    HitOffsetsOnDevice -> LocateOnDevice) run concurrently on one image -- two scratch slots, one set of sort temporaries,
    gates between the streams -- and every round of every caller must give the oracle's results"""
    import torch
    n, K = 300000, 15
    txt = synth.text(n + 5, n, synth.DNA_ALPHABET).copy()
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    g.set_deep_seed(11)
    dev = torch.device("cuda")
    jobs = []
    for c in range(callers):
        Q = 60000 + 1777 * c
        m = Q // (3 + c)
        q = np.concatenate([synth.random_queries(500 + c, Q - m, K), synth.planted_queries(510 + c, m, K, txt)])
        q = q[np.random.default_rng(c).permutation(Q)]
        chars, offsets = synth.fixed_csr(q)
        sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=4)
        has = np.flatnonzero(cnt > 0)
        oho, opos, _ = oi.batch_locate(sp[has], ep[has], threads=4)
        jobs.append({"Q": Q, "chars": chars, "sp": sp, "ep": ep, "cnt": cnt, "has": has, "oho": oho, "opos": opos, "errors": []})

    def work(job):
        try:
            Q = job["Q"]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                d_chars = torch.from_numpy(job["chars"]).to(dev)
                d_ranges = torch.zeros(Q * 2, dtype=torch.int64, device=dev)
                d_counts = torch.zeros(Q, dtype=torch.int32, device=dev)
                d_kmers = torch.zeros(Q, dtype=torch.int32, device=dev)
                d_list = torch.zeros(Q * 2, dtype=torch.int64, device=dev)
                d_num = torch.zeros(1, dtype=torch.int32, device=dev)
                d_off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
                d_scratch = torch.zeros(awfm.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
                d_pos = torch.zeros(len(job["opos"]) + 64, dtype=torch.int64, device=dev)
            stream.synchronize()
            s = stream.cuda_stream
            for _ in range(12):
                g.search_hits(d_chars.data_ptr(), 0, K, Q, d_ranges.data_ptr(), d_counts.data_ptr(), s)
                g.search_hits_compact(d_chars.data_ptr(), 0, K, Q, d_kmers.data_ptr(), d_list.data_ptr(), Q, d_num.data_ptr(), stream=s)
                g.sort_hits_on_device(d_kmers.data_ptr(), d_list.data_ptr(), Q, d_num.data_ptr(), Q, s)
                g.hit_offsets_on_device(0, d_list.data_ptr(), Q, d_off.data_ptr(), d_scratch.data_ptr(), s)
                g.locate_on_device(d_list.data_ptr(), d_off.data_ptr(), Q, d_pos.numel(), d_pos.data_ptr(), s)
                stream.synchronize()
                counts = d_counts.cpu().numpy().view(np.uint32)
                ranges = d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2)
                hit = job["cnt"] > 0
                if not np.array_equal(counts, job["cnt"]):
                    raise AssertionError("dense counts differ from the oracle")
                if not (np.array_equal(ranges[hit, 0], job["sp"][hit]) and np.array_equal(ranges[hit, 1], job["ep"][hit])):
                    raise AssertionError("dense ranges differ from the oracle")
                listed = int(d_num.item())
                if listed != len(job["has"]) or not np.array_equal(d_kmers[:listed].cpu().numpy().view(np.uint32), job["has"]):
                    raise AssertionError("the list of k-mers with hits differs from the oracle")
                if not np.array_equal(d_off[:listed + 1].cpu().numpy().view(np.uint64), job["oho"]):
                    raise AssertionError("hit offsets over the list differ from the oracle")
                if not np.array_equal(d_pos[:len(job["opos"])].cpu().numpy().view(np.uint64), job["opos"]):
                    raise AssertionError("positions differ from the oracle")
        except Exception as e:  # noqa: BLE001
            job["errors"].append(repr(e))

    threads = [threading.Thread(target=work, args=(job,)) for job in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not any(t.is_alive() for t in threads), "a caller did not come back"
    for i, job in enumerate(jobs):
        assert not job["errors"], f"caller {i}: {job['errors']}"
    g.destroy()
    ix.dealloc()


def test_mixed_length_callers_race_for_the_length_tables(oracle, awfm, require_gpu, monkeypatch):
    """three threads on their own streams send mixed-length batches to one fresh image at once: the first of them builds the
    tables per k-mer length (under the image's lock, while the others wait for it), then all three run the lookup-first
    kernel side by side -- two scratch slots, the third queues -- for several rounds; counts and hit ranges of every
    caller against the oracle"""
    import torch
    monkeypatch.setenv("AWFM_GPU_MIXED_LOOKUP", "1")
    n = 300000
    txt = synth.text(n + 9, n, synth.DNA_ALPHABET).copy()
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    g.set_deep_seed(12)
    assert g.length_tables[0] == 0
    dev = torch.device("cuda")
    jobs = []
    for c in range(3):
        Q = 50000 + 3331 * c
        chars, offsets = synth.mixed_queries(600 + c, Q, txt, synth.DNA_ALPHABET, 1 + c, 34)
        sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=4)
        jobs.append({"Q": Q, "chars": chars, "offsets": offsets, "sp": sp, "ep": ep, "cnt": cnt, "errors": []})
    start = threading.Barrier(3)

    def work(job):
        try:
            Q = job["Q"]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                d_chars = torch.from_numpy(np.concatenate([job["chars"], np.zeros(8, np.uint8)])).to(dev)
                d_off = torch.from_numpy(job["offsets"].view(np.int64)).to(dev)
                d_ranges = torch.zeros(Q * 2, dtype=torch.int64, device=dev)
                d_counts = torch.zeros(Q, dtype=torch.int32, device=dev)
            stream.synchronize()
            start.wait(60)
            for _ in range(8):
                g.search_hits(d_chars.data_ptr(), d_off.data_ptr(), 0, Q, d_ranges.data_ptr(), d_counts.data_ptr(), stream.cuda_stream)
                stream.synchronize()
                counts = d_counts.cpu().numpy().view(np.uint32)
                ranges = d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2)
                hit = job["cnt"] > 0
                if not np.array_equal(counts, job["cnt"]):
                    raise AssertionError("counts differ from the oracle")
                if not (np.array_equal(ranges[hit, 0], job["sp"][hit]) and np.array_equal(ranges[hit, 1], job["ep"][hit])):
                    raise AssertionError("ranges of the hits differ from the oracle")
                with torch.cuda.stream(stream):  # (on the caller's own stream: torch's current stream in a new thread is the null
                    d_counts.fill_(9)            # stream, which a non-blocking stream does not wait for -- the fills raced with
                    d_ranges.fill_(9)            # the next round's results)
                stream.synchronize()
        except Exception as e:  # noqa: BLE001
            job["errors"].append(repr(e))

    threads = [threading.Thread(target=work, args=(job,)) for job in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not any(t.is_alive() for t in threads), "a caller did not come back"
    for c, job in enumerate(jobs):
        assert not job["errors"], (c, job["errors"])
    assert g.length_tables[0] == 8 * (4 ** 12 - 4) // 3 and g.last_ordered_kernel_is_lookup()
    g.destroy()
    ix.dealloc()


def test_a_stream_is_retired_before_it_is_destroyed(oracle, awfm, require_gpu):
    """awfmGpuStreamRetire (round 5): the image keeps the handle of the last stream that used a scratch slot, because the event
    of that use is recorded lazily -- on that stream, when another stream shows up.  A caller that destroys a stream first
    retires it: the owed event is recorded while the handle is good.  Streams come and go between searches here, every one
    retired; the searches on the streams after them must wait for the retired streams' work and give the oracle's hits."""
    import torch
    n, K, Q = 300000, 15, 60007
    txt = synth.text(n + 11, n, synth.DNA_ALPHABET).copy()
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    g = awfm.GpuIndex(ix)
    g.set_ordered(1)
    g.set_deep_seed(11)
    dev = torch.device("cuda")
    batches = []
    for c in range(3):
        q = np.concatenate([synth.random_queries(31 + c, Q // 2, K), synth.planted_queries(41 + c, Q - Q // 2, K, txt)])
        chars, offsets = synth.fixed_csr(q)
        sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=4)
        batches.append((torch.from_numpy(chars).to(dev), sp, ep, cnt))
    d_ranges = [torch.zeros(Q * 2, dtype=torch.int64, device=dev) for _ in range(3)]
    d_counts = [torch.zeros(Q, dtype=torch.int32, device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    for round_ in range(6):
        streams = [torch.cuda.Stream() for _ in range(3)]
        for c, st in enumerate(streams):  # three streams, two scratch slots: the third queues behind a lazily recorded gate
            g.search_hits(batches[c][0].data_ptr(), 0, K, Q, d_ranges[c].data_ptr(), d_counts[c].data_ptr(), st.cuda_stream)
        for c, st in enumerate(streams):
            st.synchronize()
            _, sp, ep, cnt = batches[c]
            counts = d_counts[c].cpu().numpy().view(np.uint32)
            ranges = d_ranges[c].cpu().numpy().view(np.uint64).reshape(Q, 2)
            hit = cnt > 0
            assert np.array_equal(counts, cnt), (round_, c)
            assert np.array_equal(ranges[hit, 0], sp[hit]) and np.array_equal(ranges[hit, 1], ep[hit]), (round_, c)
            g.stream_retire(st.cuda_stream)
        del streams  # destroyed: the next round's streams may well get the same handles
        for c in range(3):
            d_counts[c].fill_(9)
            d_ranges[c].fill_(9)
        torch.cuda.synchronize()
    g.destroy()
    ix.dealloc()


def test_accelerators_built_behind_the_first_searches(oracle, awfm, require_gpu, monkeypatch):
    """Round 6: the image awFmParallelSearch* makes for an index is usable as soon as its blocks, tables and pair image are on the
    device; its deeper table and its full suffix array are built by a thread of their own and installed between two calls
    (awfm_gpu_image.hip).  Three host threads search the index from the first moment on, with their own lists, round after round:
    every round of every caller -- before, while and after the accelerators arrive -- must give the oracle's counts and position
    lists (the reference's index answers the moment it is loaded, ref src/AwFmFile.c:195-449); and the explicit way to the image
    (awfmGpuIndexAcquire) hands it over complete."""
    monkeypatch.setenv("AWFM_GPU_DEEP_SEED_K", "11")  # (a small text gets neither by itself)
    monkeypatch.setenv("AWFM_GPU_DENSE_SA", "auto")
    txt = synth.text(451, 900_000).copy()
    txt[1000:1100] = ord("n")
    ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)  # the host builder's index: no device image yet
    oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    n, jobs = 30011, []
    for t in range(3):
        q = np.concatenate([synth.random_queries(460 + t, n // 2, 19), synth.planted_queries(470 + t, n - n // 2, 19, txt)])
        kmers = [bytes(r) for r in q]
        sp, ep, cnt, _ = oi.search_list(kmers)
        hit_off, pos, _ = oi.batch_locate(sp, ep)
        lst = awfm.KmerSearchList(n)
        lst.fill(kmers)
        jobs.append({"cnt": cnt, "hit_off": hit_off, "pos": pos, "list": lst, "errors": [], "rounds": 0})

    def work(job):
        try:
            for _ in range(25):
                rc = awfm.parallel_search_locate(ix, job["list"], 2)
                if rc != awfm.AwFmSuccess:
                    raise AssertionError(f"awFmParallelSearchLocate returned {rc}")
                if not np.array_equal(job["list"].counts(), job["cnt"]):
                    raise AssertionError(f"round {job['rounds']}: counts differ from the oracle")
                for i in range(0, n, 37):
                    if not np.array_equal(job["list"].positions(i), job["pos"][int(job["hit_off"][i]):int(job["hit_off"][i + 1])]):
                        raise AssertionError(f"round {job['rounds']}: positions of k-mer {i} differ from the oracle")
                job["rounds"] += 1
        except Exception as e:  # noqa: BLE001  (reported by the main thread)
            job["errors"].append(repr(e))

    threads = [threading.Thread(target=work, args=(job,)) for job in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not any(t.is_alive() for t in threads), "a caller did not come back"
    for i, job in enumerate(jobs):
        assert not job["errors"] and job["rounds"] == 25, f"caller {i}: {job['errors']}"
    g = awfm.GpuIndex(ix, acquire=True)  # the explicit way: complete
    assert g.deep_seed_k == 11 and g.has_dense_sa, g.describe()
    awfm.parallel_search_locate(ix, jobs[0]["list"], 2)  # ... and the drop-in call through the complete image
    assert np.array_equal(jobs[0]["list"].counts(), jobs[0]["cnt"])
    for i in range(0, n, 37):
        assert np.array_equal(jobs[0]["list"].positions(i), jobs[0]["pos"][int(jobs[0]["hit_off"][i]):int(jobs[0]["hit_off"][i + 1])])
    g.handle = None
    for job in jobs:
        job["list"].dealloc()
    ix.dealloc()
