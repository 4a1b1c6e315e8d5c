#!/usr/bin/env python3
"""bench.py -- Mkmers/s located on a GRCh38-sized synthetic nucleotide FM-index, MI355X.

One "step" = one pass of the hot path over one batch of synthetic k-mers already resident in HBM:
search (awfmGpuSearchHits: seed lookup + backward search; large fixed-length batches in seed order,
DESIGN.md 4a) -> hit-offset scan -> expand + LF-walk/SA kernels, i.e. the device side of
awFmParallelSearchLocate (ref src/AwFmParallelSearch.c:95-157).

Workload (BASELINE.json configs[2]): 100 M uniform random 21-mers, locate, index of a 3.1 Gbp
uniform synthetic text, SA compression 8, seed table k=12, one index replica per GPU, the query
batch sharded over the ranks with no collective.  --scaling strong (default; configs[2] as written: "query
batch sharded 1->8"): ONE batch of --queries k-mers cut into N contiguous shards; --scaling weak: every rank
has its own 100 M k-mers of a 100 M x N batch.  A 1-GPU run also times the shards 2, 4 and 8 ranks would hold
(`scaling_proxy`).  `--workload planted` runs the secondary
case (k-mers drawn from the text, >=1 hit each), `--workload mixed` configs[4] (8..30-mers; shards
balanced by the sum of the lengths).

The JSON line also carries
  roofline         -- the dominant kernel.  Seed-order path: COMPULSORY bytes of orderedSearchKernel (distinct
                      128-B lines per search level, tallied on the device by an instrumented launch, plus records
                      read and results stored) / that kernel's HIP-event time, against the 8 TB/s HBM peak; the
                      L2-side figures and the whole call (encode + sort + search) priced by the reference
                      algorithm's bytes are beside it, the latter NOT as a roofline fraction.
                      General kernel: algorithmic bytes (L + 16 t + 104 D + 16 per query) / HIP-event time.
  roofline_general -- the exact-range general kernel (awfmGpuSearch) on the same batch: the HBM-bound kernel
                      north_star describes, priced by the algorithmic bytes;
  cpu_baseline     -- the CPU oracle (a port of the reference's OpenMP 8-query lock-step driver) timed
                      on this box's host cores on a bounded sample of the same queries, after checking
                      that its results equal the GPU's on that sample;
  digests          -- additive digests of every rank's counts / positions, checked against the committed
                      digests of the 1-rank run (tests/golden/bench_digests.json).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_GATHER_PEAK_GBS = 17800.0  # MI355X_MICROARCH.md "Indexed rows": 16.8-18.8 TB/s chip-wide for rows served by the XCDs' L2s
RANK_BYTES_DNA, RANK_BYTES_AMINO = 104, 168  # planes + one count of the reference block
PROFILE_ROUND = "r6"
T_START = time.time()
WINDOW_HITS = 1 << 28  # hits located per window when a batch's hit list is not kept resident (--workload mixed --mode locate)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--workload", choices=["random", "planted", "mixed", "unique"], default="random",
                   help="unique (with --text repetitive): k-mers drawn from the unique sequence of the genome-shaped text -- every "
                        "k-mer has a hit and few besides, so the batch can be LOCATED (planted k-mers of that text are counted: a "
                        "21-mer out of a repeat family has 10^5 hits)")
    p.add_argument("--mixed-lengths", type=int, nargs=2, default=[8, 30], metavar=("LO", "HI"),
                   help="--workload mixed: the k-mer lengths (configs[4]: 8 30)")
    p.add_argument("--mode", choices=["locate", "count"], default=None,
                   help="default: locate (mixed lengths: count; --workload mixed --mode locate is 2 M k-mers located in windows)")
    p.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                   help="strong (default; configs[2]: ONE batch of --queries k-mers, sharded over the --gpus ranks) | "
                        "weak: --queries k-mers per GPU")
    count = lambda v: int(float(v))  # noqa: E731  ("3e6" is accepted)
    p.add_argument("--seed-bucket-pipeline", type=int, default=1,
                   help="--sharding seed_bucket: 1 = two batches in flight (the exchange of one behind the search of the one before), 0 = one")
    p.add_argument("--sharding", choices=["contiguous", "seed_bucket"], default="contiguous",
                   help="how a batch is cut over the ranks: contiguous stretches of the batch (no exchange; the default), or -- dense-hit "
                        "fixed-length nucleotide batches, locate -- by seed bucket: every rank orders its stretch, the ranks exchange the records "
                        "bucket range by bucket range (one all-to-all), every rank searches a dense N-th of the ORDER (round 6)")
    p.add_argument("--text-len", type=count, default=None, help="default 3.1e9 (dna) / 2e8 (amino)")
    p.add_argument("--queries", type=count, default=None, help="k-mers per GPU per step (strong: in total); default 1e8 (dna) / 5e7 (amino)")
    p.add_argument("--query-offset", type=count, default=0,
                   help="global number of the batch's first k-mer (a 1-GPU run can stand in for a later shard: digests)")
    p.add_argument("--kmer", type=int, default=None, help="default 21 (dna) / 10 (amino)")
    p.add_argument("--seed-k", type=int, default=None, help="default 12 (dna) / 5 (amino)")
    p.add_argument("--sa-ratio", type=int, default=8)
    p.add_argument("--alphabet", choices=["dna", "amino"], default="dna")
    p.add_argument("--text", choices=["uniform", "repetitive"], default="uniform",
                   help="repetitive: a genome-shaped text (repeat families, tandem repeats, runs of N; synth.py)")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    p.add_argument("--no-cpu", action="store_true")
    p.add_argument("--no-e2e", action="store_true", help="skip the host-inclusive end_to_end leg")
    p.add_argument("--no-secondary", action="store_true", help="skip the planted (every k-mer has a hit) line of the default run")
    p.add_argument("--no-amino", action="store_true", help="skip the amino lines (configs[3]) of the default run's secondary")
    p.add_argument("--no-repetitive", action="store_true", help="skip the genome-shaped-text line of the default run's secondary")
    p.add_argument("--no-wide", action="store_true", help="skip the 6.2 Gbp index (beyond 2^32 positions) of the default run's secondary")
    p.add_argument("--no-shard-proxy", dest="shard_proxy", action="store_false",
                   help="skip the single-GPU strong-scaling proxy (the shards 2, 4 and 8 ranks would hold, each timed alone)")
    p.add_argument("--proxy-steps", type=int, default=5, help="timed steps per shard of the proxy")
    p.add_argument("--streams", type=int, default=1,
                   help="streams the steps alternate between (each with its own result buffers; the image has two scratch slots): with 2 "
                        "the launch-bound tail of step i -- ranking the list, its offsets, the locate of a few hits -- runs beside "
                        "the table gather of step i + 1 (10^8 random 21-mers 2.85 -> 2.80 ms, a 12.5 M shard 0.45 -> 0.42 ms; the "
                        "dominant kernels of two steps then overlap as well, so their own times say less).  1 (default): every "
                        "step behind the one before it")
    p.add_argument("--no-dense-form", dest="dense_form", action="store_false",
                   help="skip timing the dense form of the results beside the form the steps used")
    p.add_argument("--general-steps", type=int, default=10,
                   help="timed steps of the exact-range general kernel for roofline_general (0: skip)")
    p.add_argument("--e2e-aos-queries", type=count, default=10_000_000,
                   help="k-mers of the batch that also go through the drop-in AoS entry point (0: skip)")
    p.add_argument("--dist-backend", default="nccl", help="nccl (RCCL; falls back to gloo in-process when it cannot be set up) | gloo")
    p.add_argument("--force-device", type=int, default=-1, help="testing: every rank uses this GPU")
    p.add_argument("--device-dense-sa", action="store_true", default=None,
                   help="device-only full suffix array (same positions, a locate becomes one gather): on whatever the library's "
                        "default for the image (large images have it when memory allows)")
    p.add_argument("--no-device-dense-sa", dest="device_dense_sa", action="store_false",
                   help="locate through the LF walk and the sampled suffix array even where the image would carry the full one")
    p.add_argument("--device-seed-k", type=int, default=-1,
                   help="device-only deeper seed table (same results, fewer block reads): -1 = the library's default for "
                        "the image, 0 = none, k = that depth")
    p.add_argument("--dump-dir", default=None,
                   help="testing: every rank writes its shard's counts / hit offsets / positions to <dir>/rank<r>.npz")
    p.add_argument("--record-digests", default=None,
                   help="append this run's per-shard digests to the JSON file (how tests/golden/bench_digests.json is made)")
    return p.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without an outer torchrun: this process starts the N ranks itself.

    The parent never touches the GPU (no HIP call, no torch.cuda call before or after the spawn): it only
    starts one child per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's JSON
    line and fails if any rank failed.  Ranks are independent (index replica per GPU, contiguous query
    shards, ref src/AwFmParallelSearch.c:103-129: 8-query blocks are independent), so nothing else is shared.
    All children are polled: the first one that fails takes the others down (they would otherwise sit in the
    rendezvous until its time-out), and none outlives the parent."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    out_file = tempfile.TemporaryFile()
    try:
        for r in range(args.gpus):
            # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC; RCCL's intra-node
            # transport (hipIpcGetMemHandle) fails with "invalid argument" without it.  The image exports it already;
            # it is repeated here so that a caller's scrubbed environment does not break the N > 1 runs.
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out_file if r == 0 else subprocess.DEVNULL))
        codes = [None] * len(procs)
        failed_at = None
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
            if failed_at is None and any(c not in (None, 0) for c in codes):
                failed_at = time.time()
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        p.terminate()
            if failed_at is not None and time.time() - failed_at > 10.0:
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        p.kill()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    out_file.seek(0)
    out = out_file.read().decode()
    sys.stdout.write(out)
    sys.stdout.flush()
    if any(codes):
        sys.stderr.write(f"bench.py: rank exit codes {codes}\n")
        sys.exit(1)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if not lines or json.loads(lines[-1]).get("n_gpus") != args.gpus:
        sys.stderr.write("bench.py: rank 0 did not report n_gpus == --gpus\n")
        sys.exit(1)


def end_to_end(args, L, api, g, ix, d_chars, d_counts, d_hit_off, state, Q, K, amino, dev, d_planted=None, first=0, n=0):
    """wall-clock rates with the batch in HOST memory at the start and the results in host memory at the end"""
    import ctypes as C
    import numpy as np
    import torch
    locate = args.mode == "locate"
    out = {"unit": "Mkmers/s", "mode": args.mode}
    # the batch, bit-packed, in page-locked host memory (packed on the device from the resident ASCII batch)
    d_packed = torch.empty(Q, dtype=torch.int64, device=dev)
    bad = g.pack_device(d_chars.data_ptr(), K, Q, d_packed.data_ptr())
    assert bad == 0, "synthetic k-mers are plain letters"
    address = L.awfmGpuHostAlloc(Q * 8)
    assert address
    host_packed = np.ctypeslib.as_array(C.cast(address, C.POINTER(C.c_uint64)), shape=(Q,))
    host_packed[:] = d_packed.cpu().numpy().view(np.uint64)
    del d_packed
    seen = {"kmers": 0, "hits": 0, "chunks": 0}

    def tally_sink(user, first, m, counts, positions, total):
        seen["kmers"] += m
        seen["hits"] += total
        seen["chunks"] += 1
        return 0

    def run_stream():
        seen.update(kmers=0, hits=0, chunks=0)
        t0 = time.perf_counter()
        g.stream((address, Q), K, locate=locate, chunk=0, sink=tally_sink)
        return time.perf_counter() - t0

    run_stream()  # first call allocates the pipeline's device buffers and page-locked staging
    times = [run_stream() for _ in range(3)]
    assert seen["kmers"] == Q and (not locate or seen["hits"] == state["hits"]), "pipeline lost k-mers or hits"
    # parity at full size: the pipeline's counts / positions against the device-buffer API's results of the timed steps
    counts, positions = g.stream((address, Q), K, locate=locate, chunk=0)
    if locate:
        ho = d_hit_off.cpu().numpy().view(np.uint64)
        assert np.array_equal(counts, np.diff(ho).astype(np.uint32)), "pipeline counts differ from the device API's"
        assert np.array_equal(positions, state["positions"][: state["hits"]].cpu().numpy().view(np.uint64)), \
            "pipeline positions differ from the device API's"
    else:
        assert np.array_equal(counts, d_counts.cpu().numpy().view(np.uint32)), "pipeline counts differ from the device API's"
    del counts, positions
    best = min(times)
    out["flat_packed_pipeline"] = {
        "value": round(Q / best / 1e6, 1), "ms": round(best * 1e3, 2), "kmers": Q, "chunks": seen["chunks"],
        "entry_point": "awfmGpuStreamPacked", "input": "8 B/k-mer packed words in page-locked host memory",
        "output": "4 B/k-mer counts" + (" + 8 B/hit positions" if locate else "") + " in page-locked staging, per chunk",
        "pcie_bytes": {"h2d": Q * 8, "d2h": Q * 4 + (seen["hits"] * 8 if locate else 0)},
        "checked": "counts and positions equal the device-buffer API's for the whole batch"}
    # the same batch through the pipeline with SPARSE results (awfmGpuStreamPackedSparse): per chunk the k-mers with hits as
    # a list + positions -- the download is 12 bytes per k-mer with hits instead of 4 per k-mer
    if locate and state["listed"]:  # a batch in which few k-mers occur (the list form is what the timed steps used)
        sp_seen = {"kmers": 0, "listed": 0, "hits": 0}

        def sparse_sink(user, first, m, num, hit_kmers, hit_offsets, positions, total):
            sp_seen["kmers"] += m
            sp_seen["listed"] += num
            sp_seen["hits"] += total
            return 0

        def run_sparse():
            sp_seen.update(kmers=0, listed=0, hits=0)
            t0 = time.perf_counter()
            g.stream_sparse((address, Q), K, locate=True, chunk=0, sink=sparse_sink)
            return time.perf_counter() - t0

        run_sparse()
        sparse_times = [run_sparse() for _ in range(3)]
        assert sp_seen["kmers"] == Q and sp_seen["hits"] == state["hits"], "sparse pipeline lost k-mers or hits"
        ids, off, positions = g.stream_sparse((address, Q), K, locate=True, chunk=0)
        ho = d_hit_off.cpu().numpy().view(np.uint64)
        cnt = np.diff(ho)
        assert np.array_equal(ids, np.flatnonzero(cnt).astype(np.uint64)), "sparse pipeline lists other k-mers than the device API found"
        assert np.array_equal(np.diff(off), cnt[cnt > 0]) and np.array_equal(
            positions, state["positions"][: state["hits"]].cpu().numpy().view(np.uint64)), "sparse pipeline positions differ"
        del ids, off, positions, cnt, ho
        best_sparse = min(sparse_times)
        out["flat_packed_pipeline_sparse"] = {
            "value": round(Q / best_sparse / 1e6, 1), "ms": round(best_sparse * 1e3, 2), "kmers": Q,
            "entry_point": "awfmGpuStreamPackedSparse", "kmers_with_hits": sp_seen["listed"],
            "output": "per chunk: list of the k-mers with hits (4 B) + hit offsets (8 B) + positions (8 B/hit)",
            "pcie_bytes": {"h2d": Q * 8, "d2h": sp_seen["listed"] * 12 + sp_seen["hits"] * 8},
            "checked": "listed k-mers, hit offsets and positions equal the device-buffer API's for the whole batch"}
    L.awfmGpuHostFree(address)
    # the drop-in AoS entry point on a prefix of the batch (one kmerString pointer in, one positionList out per k-mer)
    m = min(Q, args.e2e_aos_queries)
    if m:
        chars = np.ascontiguousarray(d_chars[: m * K].cpu().numpy())
        lst = api.KmerSearchList(m)
        data = lst.ptr.contents.kmerSearchData
        arr = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_uint64)), shape=(m, 4))
        arr[:, 0] = chars.ctypes.data + np.arange(m, dtype=np.uint64) * np.uint64(K)
        arr[:, 1] = K
        lst.ptr.contents.count = m
        threads = min(32, 2 * (os.cpu_count() or 1))
        fn = (lambda: api.parallel_search_locate(ix, lst, threads)) if locate else (lambda: api.parallel_search_count(ix, lst, threads))
        fn()
        aos_times = []
        for _ in range(5):  # host threads on a shared box: the best of five calls, every call in `ms_all`
            t0 = time.perf_counter()
            fn()
            aos_times.append(time.perf_counter() - t0)
        dt = min(aos_times)
        got = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_uint32)), shape=(m, 8))[:, 6]
        if locate:
            ho = d_hit_off[: m + 1].cpu().numpy().view(np.uint64)
            assert np.array_equal(got, np.diff(ho).astype(np.uint32)), "AoS counts differ from the device API's"
        else:
            assert np.array_equal(got, d_counts[:m].cpu().numpy().view(np.uint32)), "AoS counts differ from the device API's"
        def host_side(value):
            """the host stages of the LAST call (awfmGpuAosLastStages) against what this box's memory system gives the same
            threads: pack and scatter move bytes and nothing else, so bytes moved / measured copy rate bounds a call from below"""
            st = (C.c_double * 10)()
            L.awfmGpuAosLastStages(C.byref(st))
            wall, turn, pack, device, scatter, chunks, kmers, hits, pack_b, scatter_b = list(st)
            bound_ms = (pack_b + scatter_b) / (copy_gbs * 1e9) * 1e3 if copy_gbs else None
            return {"last_call_ms": round(wall, 2), "chunks": int(chunks), "stage_ms_summed_over_chunks": {
                        "waiting_for_the_host_turn": round(turn, 2), "pack": round(pack, 2), "device_call": round(device, 2), "scatter": round(scatter, 2)},
                    "bytes_moved_by_pack": int(pack_b), "bytes_moved_by_scatter": int(scatter_b), "bytes_per_kmer": round((pack_b + scatter_b) / max(kmers, 1), 1),
                    "host_copy_GBs": round(copy_gbs, 1), "host_copy_what": f"1 GiB memcpy by the same {threads} pool threads, bytes read + written per second",
                    "host_bound_ms": round(bound_ms, 2) if bound_ms else None,
                    "host_bound_Mkmers_per_s": round(kmers / bound_ms / 1e3, 1) if bound_ms else None,
                    "frac_of_host_bound": round(bound_ms / wall, 3) if bound_ms else None,
                    "host_stages_GBs": round((pack_b + scatter_b) / ((pack + scatter) * 1e-3) / 1e9, 1) if pack + scatter > 0 else None}

        copy_gbs = float(L.awfmHostCopyGBs(threads, 1 << 30))
        out["aos_drop_in"] = {"value": round(m / dt / 1e6, 1), "ms": round(dt * 1e3, 2), "kmers": m, "host_threads": threads,
                              "ms_all": [round(t * 1e3, 2) for t in aos_times],
                              "entry_point": "awFmParallelSearchLocate" if locate else "awFmParallelSearchCount",
                              "host": host_side(m / dt / 1e6)}
        if locate and d_planted is not None:
            # the same call on k-mers drawn from the text: every one has a position list to size, fill and hand back
            from avxwindowfmindex_amd import synth
            pchars = np.ascontiguousarray(d_planted[: m * K].cpu().numpy())
            arr[:, 0] = pchars.ctypes.data + np.arange(m, dtype=np.uint64) * np.uint64(K)
            fn()
            planted_times = []
            for _ in range(3):
                t0 = time.perf_counter()
                fn()
                planted_times.append(time.perf_counter() - t0)
            got = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_uint32)), shape=(m, 8))[:, 6]
            assert got.min() >= 1, "a planted k-mer has no hit through the AoS entry point"
            planted_at = synth.planted_offsets(103, 1000, K, n, first=first)
            for i in range(1000):
                assert int(planted_at[i]) in lst.positions(i), "a planted k-mer's own offset is missing from its positionList"
            pdt = min(planted_times)
            out["aos_drop_in_planted"] = {"value": round(m / pdt / 1e6, 1), "ms": round(pdt * 1e3, 2), "kmers": m, "hits": int(got.sum()),
                                          "host_threads": threads, "ms_all": [round(t * 1e3, 2) for t in planted_times],
                                          "entry_point": "awFmParallelSearchLocate", "host": host_side(m / pdt / 1e6),
                                          "checked": "every k-mer has hits; the first 1000 position lists hold their planting offsets"}
        lst.dealloc()
    return out


def amino_leg(L, api, digest, torch, np, dev, n, Q=50_000_000, K=10, seed_k=5, sa_ratio=8, steps=5, record_digests=None):
    """BASELINE.json configs[3] beside the headline, in the driver's own run (round 5): Q random K-mers located against an amino
    index of n residues (seed 4 / 104, as `--alphabet amino`), the step a caller of the list form runs -- awfmGpuSearchHitsCompact
    + awfmGpuListLocateOnDevice --, the results checked against the CPU oracle on a sample and against the committed digests,
    the search call priced by what it executes (the basis of `--alphabet amino`'s own roofline) and the reference algorithm's
    kernel (no deeper table) timed beside it.  The index and its image live only for this leg."""
    import ctypes as C  # noqa: F401
    t0 = time.time()
    d_text = torch.empty(n, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 4, 1, None) == 1
    torch.cuda.synchronize()
    ix = api.gpu_create_index(d_text.data_ptr(), api.AwFmAlphabetAmino, sa_ratio, seed_k, on_device_length=n, device=dev.index)
    del d_text
    g = api.GpuIndex(ix, acquire=True)
    build_s = time.time() - t0
    d_chars = torch.empty(Q * K, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthRandomQueries(d_chars.data_ptr(), 0, Q, K, 104, 1, None) == 1
    cap = max(Q // 64, 1024)
    d_kmers = torch.empty(cap, dtype=torch.int32, device=dev)
    d_ranges = torch.empty(cap * 2, dtype=torch.int64, device=dev)
    d_skmers = torch.empty(cap, dtype=torch.int32, device=dev)
    d_sranges = torch.empty(cap * 2, dtype=torch.int64, device=dev)
    d_off = torch.empty(cap + 1, dtype=torch.int64, device=dev)
    d_num = torch.zeros(1, dtype=torch.int32, device=dev)
    stream_obj = torch.cuda.Stream()
    stream = stream_obj.cuda_stream
    torch.cuda.synchronize()
    # probe: how long the list is, how many hits (the buffers of the timed steps are sized by it, as a caller's are)
    g.search_hits_compact(d_chars.data_ptr(), 0, K, Q, d_kmers.data_ptr(), d_ranges.data_ptr(), cap, d_num.data_ptr(), stream=stream)
    g.list_locate_on_device(d_kmers.data_ptr(), d_ranges.data_ptr(), cap, d_num.data_ptr(), Q, d_skmers.data_ptr(), d_sranges.data_ptr(),
                            d_off.data_ptr(), 0, 0, stream)
    torch.cuda.synchronize()
    listed, hits = int(d_num.item()), int(d_off[cap].item())
    assert listed <= cap, "the amino batch is not one for the list form"
    cap = min(cap, max(1024, -(-(listed * 5 // 4) // 1024) * 1024))
    d_pos = torch.empty(hits + hits // 8 + 64, dtype=torch.int64, device=dev)

    def step(record=None):
        if record is not None:
            record[0].record(stream_obj)
        g.search_hits_compact(d_chars.data_ptr(), 0, K, Q, d_kmers.data_ptr(), d_ranges.data_ptr(), cap, d_num.data_ptr(), stream=stream)
        if record is not None:
            record[1].record(stream_obj)
        g.list_locate_on_device(d_kmers.data_ptr(), d_ranges.data_ptr(), cap, d_num.data_ptr(), Q, d_skmers.data_ptr(), d_sranges.data_ptr(),
                                d_off.data_ptr(), d_pos.numel(), d_pos.data_ptr(), stream)

    events = []
    for _ in range(2):  # warm-up steps carry the events that say what the search call alone takes
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        step(ev)
        events.append(ev)
    torch.cuda.synchronize()
    search_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
    t1 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t1) * 1e3 / steps
    looked_up = bool(g.last_ordered_kernel_is_lookup())
    front = g.last_lookup_front()
    kept = g.last_ordered_kept() if looked_up else 0  # k-mers still alive after their table entry (stepped by the lookup kernel or left to the general one)
    assert int(d_num.item()) == listed and int(d_off[cap].item()) == hits, "the list changed between the probe and the timed steps"
    # dense form of the results (outside the timed region): counts, hit offsets, positions in k-mer order
    kmers = d_skmers[:listed].to(torch.int64)
    assert listed == 0 or bool((kmers[1:] > kmers[:-1]).all()), "the hit list is not in k-mer order"
    lens = d_off[1:listed + 1] - d_off[:listed]
    counts = torch.zeros(Q, dtype=torch.int64, device=dev)
    counts[kmers] = lens
    hit_off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=hit_off[1:])
    assert int(hit_off[Q].item()) == hits
    # parity gate: the CPU oracle on the first m k-mers (ranges of the k-mers with hits, counts, positions in BWT order)
    from oracle import oracle as O
    oi = O.Index.wrap(O.AMINO, sa_ratio, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    m = min(Q, 1_000_000)
    chars = d_chars[: m * K].cpu().numpy()
    sp, ep, cnt, _ = oi.batch_search(chars, np.arange(m + 1, dtype=np.uint64) * np.uint64(K), threads=min(os.cpu_count() or 1, 16))
    ho, pos, _ = oi.batch_locate(sp, ep, threads=min(os.cpu_count() or 1, 16))
    assert np.array_equal(counts[:m].cpu().numpy().astype(np.uint32), cnt), "amino: GPU counts differ from the oracle"
    assert np.array_equal(hit_off[: m + 1].cpu().numpy().view(np.uint64), ho), "amino: GPU hit offsets differ from the oracle"
    assert np.array_equal(d_pos[: int(ho[-1])].cpu().numpy().view(np.uint64), pos), "amino: GPU positions differ from the oracle"
    in_sample = int((kmers < m).sum().item())
    gr = d_sranges[: 2 * in_sample].cpu().numpy().view(np.uint64).reshape(in_sample, 2)
    hit = cnt > 0
    assert np.array_equal(gr[:, 0], sp[hit]) and np.array_equal(gr[:, 1], ep[hit]), "amino: GPU ranges differ from the oracle"
    key = digest.key("amino", "random", "locate", n, str(K), seed_k, sa_ratio, 0, Q)
    dig = {"counts": f"{digest.counts_digest(0, counts):016x}", "positions": f"{digest.positions_digest(0, hit_off, d_pos[: max(hits, 1)]):016x}"}
    committed = digest.load_golden().get(key)
    assert committed is None or committed == dig, f"amino digests {dig} differ from the committed {committed}"
    if record_digests:
        known = json.load(open(record_digests)) if os.path.exists(record_digests) else {}
        known[key] = dig
        json.dump(known, open(record_digests, "w"), indent=1, sort_keys=True)
    del counts, hit_off, lens, kmers
    # roofline of the search call: what it executes (a 128-B line per lookup in the deeper table + 168 B per distinct block of
    # the steps behind it), tallied by an instrumented launch; and the reference algorithm's kernel (no deeper table) beside it
    tally = g.search_tally(d_chars.data_ptr(), 0, K, Q)
    alg_bytes = tally["chars"] + 16 * tally["seeded"] + RANK_BYTES_AMINO * tally["blocks"] + 16 * Q
    roofline = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "kernel_ms": round(search_ms, 3)}
    deep_k = g.deep_seed_k
    if deep_k:
        _diag_before = os.environ.get("AWFM_GPU_DIAG")
        os.environ["AWFM_GPU_DIAG"] = "tally_with_deep=1"  # the library's variable of test and diagnostics hooks
        executed = g.search_tally(d_chars.data_ptr(), 0, K, Q)
        os.environ.pop("AWFM_GPU_DIAG")
        if _diag_before is not None:
            os.environ["AWFM_GPU_DIAG"] = _diag_before
        lookups = Q - executed["seeded"] if K >= deep_k else 0
        exec_bytes = executed["chars"] + 128 * lookups + 16 * executed["seeded"] + RANK_BYTES_AMINO * executed["blocks"] + 16 * Q
        basis = "executed_reads"
        if looked_up:
            # the lookup kernel drops a k-mer at its entry when the entry's next-letter bit is clear: it executes fewer steps than
            # the general kernel the tally instruments.  What it reads at least: the characters, a line per table entry, one block
            # per k-mer still alive (`kept`; the few that go on for a second step read more), the listed results
            exec_bytes = Q * K + 128 * lookups + RANK_BYTES_AMINO * kept + 20 * listed
            basis = "executed_reads_lower_bound"
        roofline.update(kernel=("aminoLookupSearchKernel" if looked_up else "searchKernel") + f" (device-only table of depth {deep_k})",
                        basis=basis, achieved=round(exec_bytes / (search_ms * 1e-3) / 1e9, 1),
                        frac=round(exec_bytes / (search_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), executed_bytes=int(exec_bytes),
                        kmers_alive_after_the_table=int(kept) if looked_up else None,
                        algorithmic_frac_of_this_kernel=round(alg_bytes / (search_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3))
        d_counts = torch.empty(Q, dtype=torch.int32, device=dev)
        g.set_deep_seed(0)
        ev = []
        for i in range(4):  # (the first one warms up)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream_obj)
            g.search_hits(d_chars.data_ptr(), 0, K, Q, 0, d_counts.data_ptr(), stream)
            b.record(stream_obj)
            ev.append((a, b))
        torch.cuda.synchronize()
        plain_ms = float(np.mean([a.elapsed_time(b) for a, b in ev[1:]]))
        assert int((d_counts != 0).sum().item()) == listed, "the reference algorithm's kernel finds other k-mers"
        roofline["reference_algorithm"] = {"kernel": "searchKernel (no deeper table)", "bytes": int(alg_bytes), "kernel_ms": round(plain_ms, 3),
                                           "frac": round(alg_bytes / (plain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "steps": 3}
        roofline["reference_algorithm_frac"] = roofline["reference_algorithm"]["frac"]
        del d_counts
        # measured HBM bytes of the same kernel on the same batch: rocprofv3 PMC passes of `--alphabet amino [--text-len 2e9]`
        pname = {200_000_000: "amino", 2_000_000_000: "amino_2e9", 4_400_000_000: "amino_wide"}.get(n) if (Q, K, seed_k, sa_ratio) == (50_000_000, 10, 5, 8) else None
        traffic, tsrc = profile_file("traffic", pname) if pname and looked_up and not any(
            k.startswith("AWFM_GPU_") and k not in ("AWFM_GPU_TIME_ORDERED", "AWFM_GPU_DEVICE") for k in os.environ) else (None, None)
        if traffic and str(traffic.get("kernel", "")).startswith("aminoLookupSearchKernel"):
            roofline["traffic"] = traffic["hbm_bytes_per_launch"]
            roofline["traffic_source"] = f"{tsrc}: rocprofv3 PMC passes of `bench.py --alphabet amino` on another run of the same code, not measured by this run"
            roofline["hbm_frac_measured"] = round(roofline["traffic"] / (search_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    else:
        roofline.update(kernel="searchKernel", achieved=round(alg_bytes / (search_ms * 1e-3) / 1e9, 1),
                        frac=round(alg_bytes / (search_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
    out = {"workload": f"{Q / 1e6:g} M random {K}-mers, locate, {n / 1e9:g} Gres uniform synthetic amino text, SA ratio {sa_ratio}, seed table k={seed_k}",
           "value": round(Q / ms / 1e3, 2), "unit": "Mkmers/s", "ms_per_step": round(ms, 3), "steps": steps, "search_call_ms": round(search_ms, 3),
           "hits_per_step": hits, "kmers_with_hits": listed, "result_form": "list", "lookup_front": front,
           "index_build_s": round(build_s, 2), "device_image_bytes": g.device_bytes, "device_seed_k": deep_k, "device_dense_sa": bool(g.has_dense_sa),
           "roofline": roofline, "checked": f"first {m} k-mers against the CPU oracle (ranges of the k-mers with hits, counts, positions)",
           "digests": dict(dig, status="match" if committed else "unknown")}
    g.handle = None
    L.awfmGpuIndexRelease(ix.ptr)
    ix.dealloc()
    del d_chars, d_kmers, d_ranges, d_skmers, d_sranges, d_off, d_pos
    torch.cuda.empty_cache()
    return out


def repetitive_leg(L, api, digest, torch, np, dev, n=3_100_000_000, Q=100_000_000, K=21, seed_k=12, sa_ratio=8, steps=3, record_digests=None):
    """The text shape real users have, in the driver's own run (round 5): a genome-shaped 3.1 Gbp text (repeat families, tandem
    repeats, runs of N: synth.genome_text) indexed on the device, and Q K-mers drawn from its UNIQUE sequence LOCATED -- every
    k-mer has a hit at a known offset and few besides -- with the step the planted secondary runs (results in search order:
    awfmGpuSearchHitsInOrder + hit offsets + awfmGpuLocateOnDevice).  Checked: every k-mer listed once, every k-mer's own
    offset among its positions (first 10^6 of the order), the CPU oracle on a sample, committed digests."""
    t0 = time.time()
    d_text = torch.empty(n, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthGenomeText(d_text.data_ptr(), n, 2, None) == 1
    torch.cuda.synchronize()
    ix = api.gpu_create_index(d_text.data_ptr(), api.AwFmAlphabetDna, sa_ratio, seed_k, on_device_length=n, device=dev.index)
    g = api.GpuIndex(ix, acquire=True)
    build_s = time.time() - t0
    d_chars = torch.empty(Q * K, dtype=torch.uint8, device=dev)
    d_planted_at = torch.empty(Q, dtype=torch.int64, device=dev)
    assert L.awfmGpuSynthPlantedQueriesUnique(d_chars.data_ptr(), 0, Q, K, 106, d_text.data_ptr(), n, 2, d_planted_at.data_ptr(), None) == 1
    torch.cuda.synchronize()
    del d_text
    torch.cuda.empty_cache()
    assert g.search_hits_is_ordered(False, K, Q), "the batch is not one for the seed-order path"
    d_kmers = torch.empty(Q, dtype=torch.int32, device=dev)
    d_ranges = torch.empty(Q * 2, dtype=torch.int64, device=dev)
    d_off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
    d_scratch = torch.empty(api.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
    stream_obj = torch.cuda.Stream()
    stream = stream_obj.cuda_stream
    torch.cuda.synchronize()
    d_ocounts = torch.empty(Q, dtype=torch.int32, device=dev)  # (the counts in search order: the scan reads them, not the ranges)
    g.search_hits_in_order(d_chars.data_ptr(), 0, K, Q, d_kmers.data_ptr(), d_ranges.data_ptr(), stream=stream, d_order_counts=d_ocounts.data_ptr())
    g.hit_offsets_on_device(d_ocounts.data_ptr(), 0, Q, d_off.data_ptr(), d_scratch.data_ptr(), stream)
    torch.cuda.synchronize()
    hits = int(d_off[Q].item())
    d_pos = torch.empty(hits + hits // 8 + 64, dtype=torch.int64, device=dev)

    def step():
        g.search_hits_in_order(d_chars.data_ptr(), 0, K, Q, d_kmers.data_ptr(), d_ranges.data_ptr(), stream=stream, d_order_counts=d_ocounts.data_ptr())
        g.hit_offsets_on_device(d_ocounts.data_ptr(), 0, Q, d_off.data_ptr(), d_scratch.data_ptr(), stream)
        g.locate_on_device(d_ranges.data_ptr(), d_off.data_ptr(), Q, d_pos.numel(), d_pos.data_ptr(), stream)

    timing = os.environ.get("AWFM_GPU_TIME_ORDERED")
    os.environ["AWFM_GPU_TIME_ORDERED"] = "1"
    step()
    torch.cuda.synchronize()
    g.ordered_kernel_log()
    t1 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t1) * 1e3 / steps
    kernel_ms = float(np.mean([k for _, k in g.ordered_kernel_log()]))
    if timing is None:
        del os.environ["AWFM_GPU_TIME_ORDERED"]
    assert int(d_off[Q].item()) == hits
    # ---- checks ----
    kmers = d_kmers.to(torch.int64)
    assert int(torch.bincount(kmers, minlength=Q).max().item()) == 1, "a k-mer is missing from the order or listed twice"
    lens = d_off[1:] - d_off[:-1]
    assert int(lens.min().item()) >= 1, "a k-mer drawn from the text was not found"
    m = min(Q, 1_000_000)
    ho = d_off[: m + 1].cpu().numpy().view(np.uint64)
    pos = d_pos[: int(ho[m])].cpu().numpy().view(np.uint64)
    at = d_planted_at[kmers[:m]].cpu().numpy().view(np.uint64)
    cnt = np.diff(ho)
    one = cnt == 1
    assert np.array_equal(pos[ho[:-1][one]], at[one]), "a k-mer was located somewhere else than where it was taken from"
    for i in np.flatnonzero(~one)[:1000]:
        assert at[i] in pos[ho[i]:ho[i + 1]], "a k-mer's own offset is missing from its hit list"
    # the CPU oracle on the first k-mers of the batch: ranges, counts, positions in BWT order
    from oracle import oracle as O
    oi = O.Index.wrap(O.DNA, sa_ratio, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    ms_ = min(Q, 200_000)
    chars = d_chars[: ms_ * K].cpu().numpy()
    sp, ep, ocnt, _ = oi.batch_search(chars, np.arange(ms_ + 1, dtype=np.uint64) * np.uint64(K), threads=min(os.cpu_count() or 1, 16))
    oho, opos, _ = oi.batch_locate(sp, ep, threads=min(os.cpu_count() or 1, 16))
    slot = torch.empty(Q, dtype=torch.int64, device=dev)
    slot[kmers] = torch.arange(Q, dtype=torch.int64, device=dev)
    sl = slot[:ms_]
    gr = d_ranges.view(Q, 2)[sl].cpu().numpy().view(np.uint64)
    assert np.array_equal(gr[:, 0], sp) and np.array_equal(gr[:, 1], ep), "genome-shaped text: GPU ranges differ from the oracle"
    starts, ends = d_off[:-1][sl].cpu().numpy(), d_off[1:][sl].cpu().numpy()
    gpos = np.concatenate([d_pos[int(a):int(b)].cpu().numpy() for a, b in zip(starts[:2000], ends[:2000])]).view(np.uint64)
    assert np.array_equal(gpos, opos[: int(oho[2000])]), "genome-shaped text: GPU positions differ from the oracle"
    # digests of the whole batch (k-mer order): counts, and positions through the dense offsets
    counts = torch.zeros(Q, dtype=torch.int64, device=dev)
    counts[kmers] = lens
    dense_off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=dense_off[1:])
    dense_pos = torch.empty(max(hits, 1), dtype=torch.int64, device=dev)
    step_q = 1 << 24
    for b in range(0, Q, step_q):
        e = min(Q, b + step_q)
        lo, hi = int(d_off[b].item()), int(d_off[e].item())
        if hi > lo:
            shift = torch.repeat_interleave(dense_off[:-1][kmers[b:e]] - d_off[b:e], lens[b:e])
            dense_pos[shift + torch.arange(lo, hi, dtype=torch.int64, device=dev)] = d_pos[lo:hi]
    key = digest.key("dna-repetitive", "unique", "locate", n, str(K), seed_k, sa_ratio, 0, Q)
    dig = {"counts": f"{digest.counts_digest(0, counts):016x}", "positions": f"{digest.positions_digest(0, dense_off, dense_pos[: max(hits, 1)]):016x}"}
    committed = digest.load_golden().get(key)
    assert committed is None or committed == dig, f"genome-shaped text: digests {dig} differ from the committed {committed}"
    if record_digests:
        known = json.load(open(record_digests)) if os.path.exists(record_digests) else {}
        known[key] = dig
        json.dump(known, open(record_digests, "w"), indent=1, sort_keys=True)
    del counts, dense_off, dense_pos, slot, lens, kmers
    # roofline of the dominant kernel: compulsory lines (the main line's basis), tallied by the instrumented launch
    lines = g.search_hits_line_tally(d_chars.data_ptr(), 0, K, Q)
    compulsory = (128 * (lines["seed_table_lines"] + lines["deep_table_lines"] + lines["pair_level_lines"] + lines["nuc_level_lines"])
                  + lines["record_bytes_per_kmer"] * lines["ordered_kmers"] + 20 * lines["ordered_kmers"])
    out = {"workload": f"{Q / 1e6:g} M {K}-mers drawn from the unique sequence of a genome-shaped {n / 1e9:g} Gbp text (repeat families, tandem repeats, "
                       f"24 runs of N), locate, SA ratio {sa_ratio}, seed table k={seed_k}; results in search order",
           "value": round(Q / ms / 1e3, 2), "unit": "Mkmers/s", "ms_per_step": round(ms, 3), "steps": steps, "hits_per_step": hits,
           "index_build_s": round(build_s, 2), "device_image_bytes": g.device_bytes, "device_seed_k": g.deep_seed_k, "device_dense_sa": bool(g.has_dense_sa),
           "roofline": {"bound": "hbm", "kernel": "orderedSearchKernel", "basis": "compulsory_lines", "kernel_ms": round(kernel_ms, 3),
                        "compulsory_bytes": int(compulsory), "achieved": round(compulsory / (kernel_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(compulsory / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "compulsory": lines},
           "checked": f"every k-mer once in the order; the first {m} entries located where they were taken from; ranges of the first {ms_} k-mers and "
                      "positions of the first 2000 against the CPU oracle",
           "digests": dict(dig, status="match" if committed else "unknown")}
    g.handle = None
    L.awfmGpuIndexRelease(ix.ptr)
    ix.dealloc()
    del d_chars, d_planted_at, d_kmers, d_ranges, d_off, d_scratch, d_pos
    torch.cuda.empty_cache()
    return out


def wide_leg(L, api, digest, synth, torch, np, dev, n=6_200_000_000, Q=100_000_000, K=21, seed_k=12, sa_ratio=8, steps=5, record_digests=None):
    """Round 6: an index BEYOND 2^32 positions in the driver's own run -- a two-strand human genome's size, 6.2 Gbp of uniform text,
    indexed on the device (64-bit suffix sort) -- so that the line shows what the 64-bit instantiations of the fast path do (ref
    src/AwFmIndex.h:88-91, src/AwFmSuffixArray.c:12-18 are 64-bit throughout): the deeper table in its packed 8-byte form
    (sp36 | length12 | next16), lookupSearchKernel<K, NARROW = false>, the full suffix array in 40-bit entries.  Two batches with
    the steps the main line runs: Q random K-mers located in the LIST form (awfmGpuSearchHitsCompact + awfmGpuListLocateOnDevice),
    and Q K-mers drawn from the text located in SEARCH ORDER (awfmGpuSearchHitsInOrder + hit offsets + awfmGpuLocateOnDevice).
    Checked: the CPU oracle on the first k-mers of each (ranges, counts, positions in BWT order), planted k-mers at the offsets
    they were taken from, committed digests of both batches."""
    t0 = time.time()
    d_text = torch.empty(n, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 2, 0, None) == 1
    torch.cuda.synchronize()
    ix = api.gpu_create_index(d_text.data_ptr(), api.AwFmAlphabetDna, sa_ratio, seed_k, on_device_length=n, device=dev.index)
    g = api.GpuIndex(ix, acquire=True)
    build_s = time.time() - t0
    assert g.is_wide, "an image of 2^32 positions and more runs the 64-bit instantiations"
    d_chars = torch.empty(Q * K, dtype=torch.uint8, device=dev)
    d_planted = torch.empty(Q * K, dtype=torch.uint8, device=dev)
    assert L.awfmGpuSynthRandomQueries(d_chars.data_ptr(), 0, Q, K, 102, 0, None) == 1
    assert L.awfmGpuSynthPlantedQueries(d_planted.data_ptr(), 0, Q, K, 103, d_text.data_ptr(), n, None) == 1
    torch.cuda.synchronize()
    del d_text
    torch.cuda.empty_cache()
    assert g.search_hits_is_ordered(False, K, Q), "the batch is not one for the seed-order path"
    stream_obj = torch.cuda.Stream()
    stream = stream_obj.cuda_stream
    timing = os.environ.get("AWFM_GPU_TIME_ORDERED")
    os.environ["AWFM_GPU_TIME_ORDERED"] = "1"
    from oracle import oracle as O
    oi = O.Index.wrap(O.DNA, sa_ratio, seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
    threads = min(os.cpu_count() or 1, 16)
    out = {"workload": f"an index of {n / 1e9:g} Gbp ({ix.bwt_length} BWT positions: beyond 2^32), uniform text, SA ratio {sa_ratio}, seed table k={seed_k}, "
                       "built on the device; 64-bit positions throughout",
           "index_build_s": round(build_s, 2), "device_image_bytes": g.device_bytes, "device_seed_k": g.deep_seed_k,
           "device_dense_sa": bool(g.has_dense_sa), "device_image": g.describe()}

    # ---- random k-mers, the list form (the main line's timed step) ----
    cap = max(Q // 64, 1024)
    d_hit_kmers = torch.empty(cap, dtype=torch.int32, device=dev)
    d_hit_ranges = torch.empty(cap * 2, dtype=torch.int64, device=dev)
    d_sorted_kmers = torch.empty(cap, dtype=torch.int32, device=dev)
    d_sorted_ranges = torch.empty(cap * 2, dtype=torch.int64, device=dev)
    d_hit_off_c = torch.empty(cap + 1, dtype=torch.int64, device=dev)
    d_num_hits = torch.zeros(1, dtype=torch.int32, device=dev)
    g.search_hits_compact(d_chars.data_ptr(), 0, K, Q, d_hit_kmers.data_ptr(), d_hit_ranges.data_ptr(), cap, d_num_hits.data_ptr(), stream=stream)
    g.list_locate_on_device(d_hit_kmers.data_ptr(), d_hit_ranges.data_ptr(), cap, d_num_hits.data_ptr(), Q, d_sorted_kmers.data_ptr(),
                            d_sorted_ranges.data_ptr(), d_hit_off_c.data_ptr(), 0, 0, stream)
    torch.cuda.synchronize()
    listed, hits = int(d_num_hits.item()), int(d_hit_off_c[cap].item())
    assert listed <= cap, "more k-mers with hits than the list holds"
    cap = min(cap, max(1024, -(-(listed * 5 // 4) // 1024) * 1024))
    d_pos = torch.empty(hits + hits // 8 + 64, dtype=torch.int64, device=dev)

    def list_step():
        g.search_hits_compact(d_chars.data_ptr(), 0, K, Q, d_hit_kmers.data_ptr(), d_hit_ranges.data_ptr(), cap, d_num_hits.data_ptr(), stream=stream)
        g.list_locate_on_device(d_hit_kmers.data_ptr(), d_hit_ranges.data_ptr(), cap, d_num_hits.data_ptr(), Q, d_sorted_kmers.data_ptr(),
                                d_sorted_ranges.data_ptr(), d_hit_off_c.data_ptr(), d_pos.numel(), d_pos.data_ptr(), stream)

    for _ in range(3):  # (the lookup prediction settles within two searches)
        list_step()
    torch.cuda.synchronize()
    g.ordered_kernel_log()
    t1 = time.perf_counter()
    for _ in range(steps):
        list_step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t1) * 1e3 / steps
    log = g.ordered_kernel_log()
    fronts = [f for f, _ in log if f >= 0]
    kernel_ms = float(np.mean(fronts)) if fronts else float(np.mean([k for _, k in log]))
    looked_up = bool(g.last_ordered_kernel_is_lookup())
    assert int(d_num_hits.item()) == listed and int(d_hit_off_c[cap].item()) == hits
    # the oracle on the first k-mers: every listed k-mer's range and positions, and nothing listed that the oracle does not find
    ms_ = min(Q, 2_000_000)
    chars = d_chars[: ms_ * K].cpu().numpy()
    sp, ep, ocnt, _ = oi.batch_search(chars, np.arange(ms_ + 1, dtype=np.uint64) * np.uint64(K), threads=threads)
    oho, opos, _ = oi.batch_locate(sp, ep, threads=threads)
    lk = d_sorted_kmers[:listed].cpu().numpy().view(np.uint32).astype(np.int64)
    lr = d_sorted_ranges[: 2 * listed].view(listed, 2).cpu().numpy().view(np.uint64)
    lo = d_hit_off_c[: listed + 1].cpu().numpy().view(np.uint64)
    assert np.all(np.diff(lk) > 0), "the list is not in k-mer order"
    mine = lk < ms_
    want = np.flatnonzero(ocnt != 0)
    assert np.array_equal(lk[mine], want), "the list names other k-mers than the oracle finds"
    assert np.array_equal(lr[mine, 0], sp[want]) and np.array_equal(lr[mine, 1], ep[want]), "wide index: GPU ranges differ from the oracle"
    gp = d_pos[: int(lo[int(mine.sum())])].cpu().numpy().view(np.uint64)
    assert np.array_equal(gp, opos), "wide index: GPU positions differ from the oracle"
    counts = torch.zeros(Q, dtype=torch.int64, device=dev)
    counts[d_sorted_kmers[:listed].to(torch.int64)] = d_hit_off_c[1:listed + 1] - d_hit_off_c[:listed]
    dense_off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=dense_off[1:])
    key = digest.key("dna", "random", "locate", n, str(K), seed_k, sa_ratio, 0, Q)
    dig = {"counts": f"{digest.counts_digest(0, counts):016x}", "positions": f"{digest.positions_digest(0, dense_off, d_pos[: max(hits, 1)]):016x}"}
    committed = digest.load_golden().get(key)
    assert committed is None or committed == dig, f"wide index: digests {dig} differ from the committed {committed}"
    known = {key: dig}
    del counts, dense_off
    table_lines = Q  # (an upper bound that is within 1.5 % of the tally at this size: one 128-B line per k-mer's entry)
    out["random"] = {"workload": f"{Q / 1e6:g} M random {K}-mers, locate, list form (awfmGpuSearchHitsCompact + awfmGpuListLocateOnDevice)",
                     "value": round(Q / ms / 1e3, 2), "unit": "Mkmers/s", "ms_per_step": round(ms, 3), "steps": steps, "kmers_with_hits": listed,
                     "hits_per_step": hits, "lookup_first": looked_up,
                     "roofline": {"bound": "hbm", "kernel": "lookupSearchKernel<21, NARROW = false>" if looked_up else "orderedSearchKernel", "kernel_ms": round(kernel_ms, 3),
                                  "basis": "table lines (one 128-B line per k-mer's 8-byte entry) + characters; the survivors' block lines not counted: a lower bound",
                                  "bytes": int(128 * table_lines + Q * K), "achieved": round((128 * table_lines + Q * K) / (kernel_ms * 1e-3) / 1e9, 1),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round((128 * table_lines + Q * K) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "traffic": None},
                     "checked": f"the first {ms_} k-mers against the CPU oracle: which ones are listed, their ranges, every position in BWT order",
                     "digests": dict(dig, status="match" if committed else "unknown")}
    wc, wsrc = profile_file("counters", "wide") if (n, Q, K, seed_k, sa_ratio) == (6_200_000_000, 100_000_000, 21, 12, 8) and not any(
        k.startswith("AWFM_GPU_") and k not in ("AWFM_GPU_TIME_ORDERED", "AWFM_GPU_DEVICE") for k in os.environ) else (None, None)
    if wc and looked_up and str(wc.get("kernel", "")).startswith("lookupSearchKernel") and "hbm_read_bytes" in wc:
        # rocprofv3 PMC passes of `bench.py --text-len 6.2e9` (the same batch as this leg's, the same kernel) on another run
        wr = out["random"]["roofline"]
        wr["traffic"] = int(wc["hbm_read_bytes"] + wc.get("hbm_write_bytes", 0.0))
        wr["traffic_source"] = f"{wsrc}: rocprofv3 PMC passes of `bench.py --text-len 6.2e9` on another run of the same code; reads = 2 x FETCH_SIZE, writes = WRITE_SIZE"
        profiled_ms = wc.get("avg_ns_kernel_trace", 0.0) / 1e6
        if profiled_ms and abs(profiled_ms - kernel_ms) <= 0.15 * kernel_ms:
            wr["hbm_frac_measured"] = round(wr["traffic"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    del d_hit_kmers, d_hit_ranges, d_sorted_kmers, d_sorted_ranges, d_hit_off_c, d_pos, d_chars

    # ---- k-mers drawn from the text, results in search order (the planted secondary's step) ----
    d_kmers = torch.empty(Q, dtype=torch.int32, device=dev)
    d_ranges = torch.empty(Q * 2, dtype=torch.int64, device=dev)
    d_off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
    d_scratch = torch.empty(api.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
    g.search_hits_in_order(d_planted.data_ptr(), 0, K, Q, d_kmers.data_ptr(), d_ranges.data_ptr(), stream=stream)
    g.hit_offsets_on_device(0, d_ranges.data_ptr(), Q, d_off.data_ptr(), d_scratch.data_ptr(), stream)
    torch.cuda.synchronize()
    phits = int(d_off[Q].item())
    d_pos = torch.empty(phits + phits // 8 + 64, dtype=torch.int64, device=dev)

    def order_step():
        g.search_hits_in_order(d_planted.data_ptr(), 0, K, Q, d_kmers.data_ptr(), d_ranges.data_ptr(), stream=stream)
        g.hit_offsets_on_device(0, d_ranges.data_ptr(), Q, d_off.data_ptr(), d_scratch.data_ptr(), stream)
        g.locate_on_device(d_ranges.data_ptr(), d_off.data_ptr(), Q, d_pos.numel(), d_pos.data_ptr(), stream)

    order_step()
    torch.cuda.synchronize()
    g.ordered_kernel_log()
    psteps = min(steps, 3)
    t1 = time.perf_counter()
    for _ in range(psteps):
        order_step()
    torch.cuda.synchronize()
    pms = (time.perf_counter() - t1) * 1e3 / psteps
    pkernel_ms = float(np.mean([k for _, k in g.ordered_kernel_log()]))
    assert int(d_off[Q].item()) == phits
    kmers = d_kmers.to(torch.int64)
    assert int(torch.bincount(kmers, minlength=Q).max().item()) == 1, "a k-mer is missing from the order or listed twice"
    lens = d_off[1:] - d_off[:-1]
    assert int(lens.min().item()) >= 1, "a k-mer drawn from the text was not found"
    m = min(Q, 1_000_000)
    slot = torch.empty(Q, dtype=torch.int64, device=dev)
    slot[kmers] = torch.arange(Q, dtype=torch.int64, device=dev)
    sl = slot[:m]
    planted_at = synth.planted_offsets(103, m, K, n, first=0)
    starts, cnts = d_off[:-1][sl].cpu().numpy(), lens[sl].cpu().numpy()
    firsts = d_pos[d_off[:-1][sl]].cpu().numpy().view(np.uint64)
    one = cnts == 1
    assert np.array_equal(firsts[one], planted_at[one]), "a k-mer drawn from the text was located somewhere else"
    for i in np.flatnonzero(~one)[:1000]:
        assert planted_at[i] in d_pos[int(starts[i]): int(starts[i]) + int(cnts[i])].cpu().numpy().view(np.uint64), "a k-mer's own offset is missing from its hit list"
    mo = min(Q, 200_000)
    chars = d_planted[: mo * K].cpu().numpy()
    sp, ep, ocnt, _ = oi.batch_search(chars, np.arange(mo + 1, dtype=np.uint64) * np.uint64(K), threads=threads)
    oho, opos, _ = oi.batch_locate(sp, ep, threads=threads)
    gr = d_ranges.view(Q, 2)[slot[:mo]].cpu().numpy().view(np.uint64)
    assert np.array_equal(gr[:, 0], sp) and np.array_equal(gr[:, 1], ep), "wide index, planted: GPU ranges differ from the oracle"
    gpos = np.concatenate([d_pos[int(a): int(a) + int(c)].cpu().numpy() for a, c in zip(starts[:2000], cnts[:2000])]).view(np.uint64)
    assert np.array_equal(gpos, opos[: int(oho[2000])]), "wide index, planted: GPU positions differ from the oracle"
    counts = torch.zeros(Q, dtype=torch.int64, device=dev)
    counts[kmers] = lens
    dense_off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=dense_off[1:])
    dense_pos = torch.empty(max(phits, 1), dtype=torch.int64, device=dev)
    step_q = 1 << 24
    for b in range(0, Q, step_q):
        e = min(Q, b + step_q)
        lo_, hi_ = int(d_off[b].item()), int(d_off[e].item())
        if hi_ > lo_:
            shift = torch.repeat_interleave(dense_off[:-1][kmers[b:e]] - d_off[b:e], lens[b:e])
            dense_pos[shift + torch.arange(lo_, hi_, dtype=torch.int64, device=dev)] = d_pos[lo_:hi_]
    pkey = digest.key("dna", "planted", "locate", n, str(K), seed_k, sa_ratio, 0, Q)
    pdig = {"counts": f"{digest.counts_digest(0, counts):016x}", "positions": f"{digest.positions_digest(0, dense_off, dense_pos[: max(phits, 1)]):016x}"}
    pcommitted = digest.load_golden().get(pkey)
    assert pcommitted is None or pcommitted == pdig, f"wide index, planted: digests {pdig} differ from the committed {pcommitted}"
    known[pkey] = pdig
    out["planted"] = {"workload": f"{Q / 1e6:g} M {K}-mers drawn from the text (every k-mer has >= 1 hit), locate, results in search order",
                      "value": round(Q / pms / 1e3, 2), "unit": "Mkmers/s", "ms_per_step": round(pms, 3), "steps": psteps, "hits_per_step": phits,
                      "kernel": "orderedSearchKernel<..., NARROW = false>", "kernel_ms": round(pkernel_ms, 3),
                      "checked": f"every k-mer once in the order; the first {m} k-mers located where they were taken from; ranges of the first {mo} and "
                                 "positions of the first 2000 against the CPU oracle",
                      "digests": dict(pdig, status="match" if pcommitted else "unknown")}
    out["value"], out["value_present_kmers"], out["unit"] = out["random"]["value"], out["planted"]["value"], "Mkmers/s"
    if record_digests:
        have = json.load(open(record_digests)) if os.path.exists(record_digests) else {}
        have.update(known)
        json.dump(have, open(record_digests, "w"), indent=1, sort_keys=True)
    if timing is None:
        del os.environ["AWFM_GPU_TIME_ORDERED"]
    g.handle = None
    L.awfmGpuIndexRelease(ix.ptr)
    ix.dealloc()
    del d_planted, d_kmers, d_ranges, d_off, d_scratch, d_pos, counts, dense_off, dense_pos, slot, lens, kmers
    torch.cuda.empty_cache()
    return out


def gpu_state(card=0):
    """what the device reports of itself through sysfs -- engine / memory / fabric clock levels in use, compute and memory
    partition modes, power cap -- without a child process (a process that has initialised the GPU must not start one)"""
    import glob
    out = {}
    try:
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        if not cards:
            return None
        base = os.path.dirname(cards[min(card, len(cards) - 1)])
        for name in ("sclk", "mclk", "fclk", "socclk"):
            path = os.path.join(base, f"pp_dpm_{name}")
            if os.path.exists(path):
                cur = [ln for ln in open(path).read().split("\n") if ln.strip().endswith("*")]
                out[name] = cur[0].split(":")[1].replace("*", "").strip() if cur else None
        for name in ("current_compute_partition", "current_memory_partition", "gpu_busy_percent", "mem_busy_percent"):
            path = os.path.join(base, name)
            if os.path.exists(path):
                out[name] = open(path).read().strip()
        for hw in glob.glob(os.path.join(base, "hwmon", "hwmon*")):
            for name in ("power1_average", "power1_input", "power1_cap", "temp1_input", "temp3_input"):
                path = os.path.join(hw, name)
                if os.path.exists(path):
                    try:
                        out[name] = int(open(path).read().strip())
                    except (OSError, ValueError):
                        pass
    except OSError:
        return out or None
    return out or None


def profile_file(kind, name):
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND, f"{kind}_{name}.json")
    return (json.load(open(path)), f"profiles/{PROFILE_ROUND}/{kind}_{name}.json") if os.path.exists(path) else (None, None)


def seed_bucket_run(args, L, api, digest, shard, torch, np, dev, g, ix, d_chars, Q, first, batch_total, K, n, rank, world, build_s):
    """--sharding seed_bucket: the timed step of a rank is  order my contiguous stretch (awfmGpuOrderKmers)  ->  one all-to-all of
    the records by bucket range (dist.bucket_exchange: RCCL over xGMI under the nccl backend; through host memory under gloo, the
    tests' way)  ->  search the dense N-th of the order I then hold, in search order (awfmGpuSearchOrderedRecords)  ->  hit offsets
    ->  positions; the k-mers with ambiguity characters stay with the rank that holds their characters
    (awfmGpuSearchGeneralRecords).  One host wait per step: the slice sizes of the exchange.  Results carry the k-mers' numbers in
    the WHOLE batch; the ranks' keyed digests add up to the digest of the batch however it was cut."""
    import torch.distributed as dist
    buckets = g.order_buckets(K, batch_total)
    assert buckets, "--sharding seed_bucket: fixed-length nucleotide batches whose 8-byte records fit"
    cuts = shard.bucket_cuts(buckets, world)
    via_host = world > 1 and shard.timing_backend() != "nccl"
    # Two batches in flight (round 6): the FRONT of batch i + 1 -- order, the host's wait for the slice sizes, the exchange -- is
    # issued behind the BACK of batch i -- search, hit offsets, positions -- on a stream of its own, so that the all-to-all (RCCL
    # moves it with copy engines and a few CUs) runs while the search kernels have the chip.  --seed-bucket-pipeline 0: one batch
    # at a time (front, then back, then the next front).
    pipelined = bool(args.seed_bucket_pipeline)
    front_obj, back_obj = torch.cuda.Stream(), torch.cuda.Stream()
    sf, sb = front_obj.cuda_stream, back_obj.cuda_stream

    def new_slot():
        return {"recs": torch.empty(Q, dtype=torch.int64, device=dev), "bs": torch.empty(buckets + 3, dtype=torch.int32, device=dev),
                "gk": torch.empty(Q, dtype=torch.int32, device=dev),  # my own stretch's order: its tail are the general kernel's k-mers
                "gr": torch.empty(Q * 2, dtype=torch.int64, device=dev), "m": None, "p": None, "front_done": torch.cuda.Event(), "back_done": None}

    slots = [new_slot() for _ in range(2 if pipelined else 1)]

    def front(s):
        if s["back_done"] is not None:
            front_obj.wait_event(s["back_done"])  # the slot's arrays are still being searched by the batch before last
        g.order_kmers(d_chars.data_ptr(), K, Q, first, args.query_offset + batch_total, s["recs"].data_ptr(), s["bs"].data_ptr(), sf)
        g.search_general_records(d_chars.data_ptr(), K, Q, first, args.query_offset + batch_total, s["recs"].data_ptr(), s["bs"].data_ptr(),
                                 s["gk"].data_ptr(), s["gr"].data_ptr(), sf)
        front_obj.synchronize()  # the host wait of a step: the slice sizes of the exchange
        bs = s["bs"].cpu().to(torch.int64)
        with torch.cuda.stream(front_obj):
            if via_host:  # (the tests' way: gloo moves host tensors; the slices are put in bucket order by torch on the host)
                mine, mstart = shard.bucket_exchange(s["recs"].cpu(), bs[: buckets + 1], buckets, world, rank)
                mine = mine.to(dev)
                full = shard.full_bucket_start(mstart, cuts[rank], cuts[rank + 1], buckets).to(dev)
            else:  # records stay on the device: all-to-all over RCCL (one rank: none), awfmGpuMergeBucketRuns puts the slices together
                mine, full, s["exchange_keep"] = shard.bucket_exchange_on_device(g, s["recs"], bs[: buckets + 1], buckets, world, rank,
                                                                                 group=shard.timing_group(), stream=sf)
            m = mine.numel()
            if s["m"] != m:
                s.update(m=m, k=torch.empty(max(m, 1), dtype=torch.int32, device=dev), r=torch.empty(max(m, 1) * 2, dtype=torch.int64, device=dev),
                         o=torch.zeros(max(m, 1) + 1, dtype=torch.int64, device=dev),
                         sc=torch.empty(api.GpuIndex.scan_scratch_bytes(max(m, 1)), dtype=torch.uint8, device=dev), p=None)
                front_obj.synchronize()  # (the allocator's: these arrays are first used on the other stream)
        s["front_done"].record(front_obj)
        s["left"] = int(bs[buckets])  # entries [left, Q) of gk / gr: my general k-mers
        s["keep"] = (mine, full)

    def back(s):
        back_obj.wait_event(s["front_done"])
        mine, full = s["keep"]
        m = s["m"]
        if m:
            narrow = ix.bwt_length < (1 << 32)  # (32-bit counts are exact: the scan reads them instead of the ranges)
            if narrow and s.get("c") is None or (narrow and s["c"].numel() < m):
                s["c"] = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
            g.search_ordered_records(mine.data_ptr(), full.data_ptr(), cuts[rank], cuts[rank + 1], K, args.query_offset + batch_total,
                                     s["k"].data_ptr(), s["r"].data_ptr(), sb, d_order_counts=s["c"].data_ptr() if narrow else 0)
            if narrow:
                g.hit_offsets_on_device(s["c"].data_ptr(), 0, m, s["o"].data_ptr(), s["sc"].data_ptr(), sb)
            else:
                g.hit_offsets_on_device(0, s["r"].data_ptr(), m, s["o"].data_ptr(), s["sc"].data_ptr(), sb)
            if s["p"] is None:
                back_obj.synchronize()
                hits = int(s["o"][m].item())
                s["p"] = torch.empty(hits + hits // 8 + 64, dtype=torch.int64, device=dev)
            g.locate_on_device(s["r"].data_ptr(), s["o"].data_ptr(), m, s["p"].numel(), s["p"].data_ptr(), sb)
        s["back_done"] = torch.cuda.Event()
        s["back_done"].record(back_obj)

    def run(steps):
        """`steps` batches through the two stages; returns the slot of the last one"""
        if not pipelined:
            for _ in range(steps):
                front(slots[0])
                back(slots[0])
            return slots[0]
        front(slots[0])
        for i in range(steps):
            back(slots[i % 2])
            if i + 1 < steps:
                front(slots[(i + 1) % 2])
        return slots[(steps - 1) % 2]

    def barrier():
        shard.barrier(world, torch.cuda.synchronize)

    run(max(args.warmup, 2))
    barrier()
    t0 = time.perf_counter()
    state = run(args.steps)
    barrier()
    dt = shard.max_over_ranks((time.perf_counter() - t0) / args.steps, world, dev)
    # digests keyed by the k-mers' numbers in the whole batch: my share of the order + my own general k-mers
    m, left = state["m"], state["left"]
    dc = dp = 0
    hits = 0
    if m:
        ids = state["k"][:m].to(torch.int64)
        lens = state["o"][1:m + 1] - state["o"][:m]
        hits = int(state["o"][m].item())
        dc += digest.counts_digest_keyed(ids, lens)
        dp += digest.positions_digest_keyed(ids, state["o"][: m + 1], state["p"][: max(hits, 1)])
    if left < Q:  # (located here too: ranges -> offsets -> positions over the tail)
        gm = Q - left
        gr = state["gr"][2 * left:].contiguous()
        go = torch.zeros(gm + 1, dtype=torch.int64, device=dev)
        gsc = torch.empty(api.GpuIndex.scan_scratch_bytes(gm), dtype=torch.uint8, device=dev)
        total = g.hit_offsets(gr.data_ptr(), gm, go.data_ptr(), gsc.data_ptr())
        gp = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
        g.locate(gr.data_ptr(), go.data_ptr(), gm, total, gp.data_ptr())
        torch.cuda.synchronize()
        gids = state["gk"][left:].to(torch.int64)
        dc += digest.counts_digest_keyed(gids, go[1:] - go[:-1])
        dp += digest.positions_digest_keyed(gids, go, gp)
        hits += total
    parts = shard.gather_objects((rank, m + (Q - left), hits, dc & digest.MASK, dp & digest.MASK), world)
    if rank == 0:
        total_c = sum(p[3] for p in parts) & digest.MASK
        total_p = sum(p[4] for p in parts) & digest.MASK
        assert sum(p[1] for p in parts) == batch_total, "the ranks' shares of the order do not add up to the batch"
        key = digest.key(args.alphabet, args.workload, args.mode, n, str(K), args.seed_k, args.sa_ratio, args.query_offset, batch_total)
        dig = {"counts": f"{total_c:016x}", "positions": f"{total_p:016x}"}
        committed = digest.load_golden().get(key)
        assert committed is None or committed == dig, f"seed-bucket sharding: digests {dig} differ from the committed {committed}"
        if args.record_digests:
            known = json.load(open(args.record_digests)) if os.path.exists(args.record_digests) else {}
            known[key] = dig
            json.dump(known, open(args.record_digests, "w"), indent=1, sort_keys=True)
        out = {"metric": "Mkmers/sec located, GRCh38 nucleotide index", "value": round(batch_total / dt / 1e6, 2), "unit": "Mkmers/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "u64", "data": "synthetic",
               "config": {"workload": f"{batch_total / 1e6:g} M {args.workload} {K}-mers in total, locate, {n / 1e9:g} Gbp uniform synthetic dna text, SA ratio {args.sa_ratio}, "
                                      f"seed table k={args.seed_k}",
                          "sharding": "seed_bucket", "parallelism": f"index replica per GPU; every rank orders its stretch of the batch, ONE all-to-all of the records by "
                                                                    f"bucket range ({'through host memory over gloo' if via_host else 'RCCL' if world > 1 else 'one rank: no exchange'}), "
                                                                    "every rank searches a dense N-th of the seed order",
                          "batch_kmers": batch_total, "buckets": buckets, "index_build_s": round(build_s, 2), "timing_collective": shard.timing_backend() or "none (one rank)",
                          "batches_in_flight": 2 if pipelined else 1, "host_waits_per_step": 1},
               "roofline": None, "cpu_baseline": None,
               "digests": dict(dig, status="match" if committed else "unknown", shards=world,
                               per_rank=[{"rank": p[0], "kmers": p[1], "hits": p[2]} for p in parts])}
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)  # before anything that could initialise the GPU in this process
    import numpy as np
    import torch
    import torch.distributed as dist
    from avxwindowfmindex_amd import _lib, api, digest
    from avxwindowfmindex_amd import dist as shard

    if args.force_device >= 0:
        os.environ["LOCAL_RANK"] = str(args.force_device)
    rank, world = shard.init(args.dist_backend)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: the line would misreport n_gpus")
    _, _, local_rank = shard.env_world()
    torch.cuda.set_device(local_rank if world > 1 else max(args.force_device, 0))
    dev = torch.device("cuda", torch.cuda.current_device())
    L = _lib.lib()
    amino = args.alphabet == "amino"
    alpha = api.AwFmAlphabetAmino if amino else api.AwFmAlphabetDna
    # BASELINE.json configs[2] (dna, the headline) / configs[3] (amino, Swiss-Prot-sized)
    for name, dna_default, amino_default in (("text_len", 3_100_000_000, 200_000_000), ("queries", 100_000_000, 50_000_000),
                                             ("kmer", 21, 10), ("seed_k", 12, 5)):
        if getattr(args, name) is None:
            setattr(args, name, amino_default if amino else dna_default)
    n, K = args.text_len, args.kmer
    if args.mode is None:
        args.mode = "count" if args.workload == "mixed" else "locate"
    if args.text == "repetitive" and args.workload == "planted":
        args.mode = "count"  # a 21-mer out of a 300-character family with 10^6 copies has ~10^5 hits: 10^8 of them have 10^12
    if args.workload == "mixed" and args.mode == "locate" and args.queries == 100_000_000 and not amino and args.mixed_lengths[0] < 14:
        # 8..11-mers have 10^3..10^5 hits each (2.7 * 10^3 per k-mer on average): the hit list of 10^8 mixed k-mers is
        # 2 * 10^11 positions.  The locate form of this workload is 2 M k-mers (5 * 10^9 hits), located in windows.
        # (--mixed-lengths 14 30 and longer: a hit or a few per k-mer drawn from the text -- the whole 10^8 are located)
        args.queries = 2_000_000
    text_seed = 4 if amino else 2
    query_seed = 104 if amino else {"random": 102, "planted": 103, "mixed": 105, "unique": 106}[args.workload]
    assert args.workload != "unique" or args.text == "repetitive", "--workload unique draws from a genome-shaped text (--text repetitive)"

    # ---- index replica on this GPU (text generated and indexed on the device) ----
    t0 = time.time()
    d_text = torch.empty(n, dtype=torch.uint8, device=dev)
    if args.text == "repetitive":
        assert not amino, "--text repetitive is a nucleotide text"
        assert L.awfmGpuSynthGenomeText(d_text.data_ptr(), n, text_seed, None) == 1
    else:
        assert L.awfmGpuSynthText(d_text.data_ptr(), 0, n, text_seed, int(amino), None) == 1
    torch.cuda.synchronize()
    ix = api.gpu_create_index(d_text.data_ptr(), alpha, args.sa_ratio, args.seed_k, on_device_length=n,
                              device=dev.index)
    g = api.GpuIndex(ix, acquire=True)
    build_s = time.time() - t0
    deep_first_build = g.deep_seed_build  # (seconds, transient bytes) of the table the library built by itself, if it did
    deep_first_alloc_s = g.deep_seed_alloc_s  # ... of which inside hipMalloc
    deep_s = 0.0
    if args.device_seed_k >= 0:
        t1 = time.time()
        g.set_deep_seed(args.device_seed_k)
        torch.cuda.synchronize()
        deep_s = time.time() - t1
    dense_s = g.dense_sa_build_s if g.has_dense_sa else 0.0  # the library's own choice for the image
    if args.device_dense_sa is not None and bool(args.device_dense_sa) != g.has_dense_sa:
        t1 = time.time()
        g.set_dense_sa(bool(args.device_dense_sa))
        torch.cuda.synchronize()
        dense_s = time.time() - t1 if args.device_dense_sa else 0.0
    dense_sa_default = g.has_dense_sa  # what the timed steps locate with

    # ---- this rank's query shard, resident in HBM ----
    # weak: the global batch is --queries x world k-mers and rank r has k-mers [r Q, (r+1) Q); strong: the global batch
    # is --queries k-mers cut into `world` contiguous shards (mixed lengths: cut where the running sum of the lengths
    # passes r/world of the total, so the shards carry equal work)
    if args.scaling == "weak":
        Q, batch_total = args.queries, args.queries * world
        first = args.query_offset + rank * Q
    else:
        batch_total = args.queries
        if args.workload == "mixed" and world > 1:
            d_all = torch.empty(batch_total, dtype=torch.int64, device=dev)
            assert L.awfmGpuSynthMixedLengths(d_all.data_ptr(), args.query_offset, batch_total, args.mixed_lengths[0], args.mixed_lengths[1], query_seed, None) == 1
            prefix = torch.zeros(batch_total + 1, dtype=torch.int64, device=dev)
            torch.cumsum(d_all, 0, out=prefix[1:])
            begin, end = shard.balanced_bounds(prefix, world, rank)
            del d_all, prefix
        else:
            begin, end = shard.shard_bounds(batch_total, world, rank)
        Q, first = end - begin, args.query_offset + begin
    assert Q > 0, "an empty shard"
    d_offsets = None  # CSR offsets (mixed lengths only); fixed-length batches pass the length instead
    if args.workload == "mixed":
        d_len = torch.empty(Q, dtype=torch.int64, device=dev)
        assert L.awfmGpuSynthMixedLengths(d_len.data_ptr(), first, Q, args.mixed_lengths[0], args.mixed_lengths[1], query_seed, None) == 1
        d_offsets = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
        torch.cumsum(d_len, 0, out=d_offsets[1:])
        total_chars = int(d_offsets[-1].item())
        del d_len
        d_chars = torch.empty(total_chars, dtype=torch.uint8, device=dev)
        assert L.awfmGpuSynthMixedQueries(d_chars.data_ptr(), d_offsets.data_ptr(), first, Q, query_seed,
                                          d_text.data_ptr(), n, int(amino), None) == 1
        K = 0
    else:
        d_chars = torch.empty(Q * K, dtype=torch.uint8, device=dev)
    if args.workload == "mixed":
        pass
    elif args.workload == "random":
        assert L.awfmGpuSynthRandomQueries(d_chars.data_ptr(), first, Q, K, query_seed, int(amino), None) == 1
    elif args.workload == "unique":
        assert L.awfmGpuSynthPlantedQueriesUnique(d_chars.data_ptr(), first, Q, K, query_seed, d_text.data_ptr(), n, text_seed, None, None) == 1
    else:
        # k-mers drawn from a text with runs of N get their N replaced: a k-mer of N matches every window of every run
        plant = L.awfmGpuSynthPlantedQueriesClean if args.text == "repetitive" else L.awfmGpuSynthPlantedQueries
        assert plant(d_chars.data_ptr(), first, Q, K, query_seed, d_text.data_ptr(), n, None) == 1
    torch.cuda.synchronize()
    if args.sharding == "seed_bucket":
        assert not amino and d_offsets is None and args.mode == "locate" and args.scaling == "strong", \
            "--sharding seed_bucket: fixed-length nucleotide batches, located, one batch cut over the ranks"
        return seed_bucket_run(args, L, api, digest, shard, torch, np, dev, g, ix, d_chars, Q, first, batch_total, K, n, rank, world, build_s)
    off_ptr = d_offsets.data_ptr() if d_offsets is not None else 0
    # the default run also reports the dense-hit case (every k-mer located, ~7 LF steps per hit) beside the headline:
    # its k-mers are drawn from the text now, searched after everything else
    d_planted = None
    if (args.workload == "random" and not amino and world == 1 and not args.no_secondary and args.mode == "locate"
            and args.text == "uniform"):
        d_planted = torch.empty(Q * K, dtype=torch.uint8, device=dev)
        assert L.awfmGpuSynthPlantedQueries(d_planted.data_ptr(), first, Q, K, 103, d_text.data_ptr(), n, None) == 1
        torch.cuda.synchronize()
    # ... and cfg 5's batch (configs[4]: k-mers of 8..30 characters, every second one drawn from the text; counted)
    d_mixed = None
    if d_planted is not None and n >= (1 << 28):
        m_len = torch.empty(Q, dtype=torch.int64, device=dev)
        assert L.awfmGpuSynthMixedLengths(m_len.data_ptr(), first, Q, 8, 30, 105, None) == 1
        m_off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
        torch.cumsum(m_len, 0, out=m_off[1:])
        del m_len
        m_chars = torch.empty(int(m_off[-1].item()), dtype=torch.uint8, device=dev)
        assert L.awfmGpuSynthMixedQueries(m_chars.data_ptr(), m_off.data_ptr(), first, Q, 105, d_text.data_ptr(), n, 0, None) == 1
        torch.cuda.synchronize()
        d_mixed = (m_chars, m_off)
    del d_text  # planted k-mers are already copied out; free 3.1 GB
    torch.cuda.empty_cache()

    d_ranges = torch.empty(Q * 2, dtype=torch.int64, device=dev)
    d_counts = torch.empty(Q, dtype=torch.int32, device=dev)
    d_hit_off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
    d_scratch = torch.empty(api.GpuIndex.scan_scratch_bytes(Q), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    narrow_counts = ix.bwt_length < (1 << 32)  # 32-bit counts are exact: hit offsets can be scanned from them
    ordered = g.search_hits_is_ordered(d_offsets is not None, K, Q)
    if ordered:
        os.environ["AWFM_GPU_TIME_ORDERED"] = "1"  # HIP events around the dominant kernel inside the library, logged per search
    locate = args.mode == "locate"
    dense_only = bool(os.environ.get("AWFM_BENCH_DENSE_RESULTS"))

    # ---- the forms a step's results can take (config.result_format says which one the timed steps used) ----
    # "list"  (awfmGpuSearchHitsCompact): when few k-mers of a batch occur -- 7 * 10^4 of 10^8 random 21-mers -- the k-mers with
    #         hits are a LIST {k-mer number, range}, in k-mer order, with hit offsets over that list: nothing of size 10^8 is
    #         filled, scanned or expanded after the search.  Same information as the dense form (a k-mer not listed has count 0).
    # "order" (awfmGpuSearchHitsInOrder): when most k-mers have hits, every k-mer gets an entry {k-mer number, range} in the
    #         order the seed-order search took it -- whole-line stores instead of 10^8 partial-line ones under the original
    #         numbers -- and hit offsets and positions follow that order.  A consumer that scatters into per-k-mer lists (the
    #         AoS API) reads it as it is.
    # "dense" range / count under every k-mer number + hit offsets over the batch + positions in k-mer order: what the
    #         reference's own result is, flattened.  `config.dense_form` times it beside whichever form the steps used.
    # Which one: decided by an untimed PROBE step per batch (it reads the number of listed k-mers and the number of hits back
    # and sizes the buffers).  The timed steps then run WITHOUT ONE HOST WAIT: the list's length and the hit total are read
    # on the device (awfmGpuSortHitsOnDevice / awfmGpuHitOffsetsOnDevice / awfmGpuLocateOnDevice), and what they were is
    # checked against the probe's after the timed region.
    list_cap = max(Q // 64, 1024)
    # (amino: large fixed-length batches whose k-mers reach the device-only deeper table are looked up first, and that kernel
    # can list its hits as the seed-order search does: awfm_amino_lookup_kernel.h)
    amino_lookup = bool(amino and d_offsets is None and g.deep_seed_k and g.deep_seed_k <= K <= 19 and K - g.deep_seed_k <= 12
                        and os.environ.get("AWFM_GPU_AMINO_LOOKUP") != "0")
    have_list = (ordered or amino_lookup) and locate and not dense_only
    have_order = ordered and locate and not dense_only
    if have_list:
        d_hit_kmers = torch.empty(list_cap, dtype=torch.int32, device=dev)
        d_hit_ranges = torch.empty(list_cap * 2, dtype=torch.int64, device=dev)
        d_hit_off_c = torch.empty(list_cap + 1, dtype=torch.int64, device=dev)
        d_num_hits = torch.zeros(1, dtype=torch.int32, device=dev)
        # the list as the search appends it (d_hit_kmers / d_hit_ranges) and in k-mer order (what every check reads):
        # awfmGpuListLocateOnDevice makes the second from the first, with the hit offsets and the positions, in one launch
        d_sorted_kmers = torch.empty(list_cap, dtype=torch.int32, device=dev)
        d_sorted_ranges = torch.empty(list_cap * 2, dtype=torch.int64, device=dev)
    d_order_kmers = torch.empty(Q, dtype=torch.int32, device=dev) if have_order else None
    pos_buf = {"t": None}

    class Lane:
        """the result buffers and the stream of one step in flight.  Lane 0 is the buffers above on the current stream (every
        check reads them); the others exist so that consecutive steps overlap: a step's tail is a dozen launches of kernels
        that leave the chip idle, and the next step's table gather does not depend on it."""

        def __init__(self, primary):
            self.primary = primary
            if primary:
                self.ranges, self.counts, self.hit_off, self.scratch, self.order_kmers = d_ranges, d_counts, d_hit_off, d_scratch, d_order_kmers
                if have_list:
                    self.hit_kmers, self.hit_ranges, self.hit_off_c, self.num_hits = d_hit_kmers, d_hit_ranges, d_hit_off_c, d_num_hits
                    self.sorted_kmers, self.sorted_ranges = d_sorted_kmers, d_sorted_ranges
            else:
                self.ranges, self.counts, self.hit_off = torch.empty_like(d_ranges), torch.empty_like(d_counts), torch.empty_like(d_hit_off)
                self.scratch = torch.empty_like(d_scratch)
                self.order_kmers = torch.empty_like(d_order_kmers) if have_order else None
                if have_list:
                    self.hit_kmers, self.hit_ranges = torch.empty_like(d_hit_kmers), torch.empty_like(d_hit_ranges)
                    self.hit_off_c, self.num_hits = torch.empty_like(d_hit_off_c), torch.zeros_like(d_num_hits)
                    self.sorted_kmers, self.sorted_ranges = torch.empty_like(d_sorted_kmers), torch.empty_like(d_sorted_ranges)
            # a stream of its own for every lane, lane 0 included: torch's current stream is the null stream, which waits for
            # every other one and which the library cannot tell apart from another thread's (it records an event per call on it)
            self.torch_stream = torch.cuda.Stream()
            self.stream = self.torch_stream.cuda_stream
            self.pos = None

    lanes = [Lane(True)]

    def ensure_positions(total):
        if pos_buf["t"] is None or pos_buf["t"].numel() < max(total, 1):
            pos_buf["t"] = None
            for ln in lanes:
                ln.pos = None
            pos_buf["t"] = torch.empty(max(total, 1) + total // 8 + 64, dtype=torch.int64, device=dev)
        for ln in lanes:
            if ln.primary:
                ln.pos = pos_buf["t"]
            elif ln.pos is None or ln.pos.numel() != pos_buf["t"].numel():
                ln.pos = torch.empty_like(pos_buf["t"])
        return pos_buf["t"]

    class Piece:
        """a contiguous piece [begin, begin + q) of this rank's k-mers: what one step searches (the whole shard, or -- for
        --shard-proxy -- the part of it one of N ranks would hold)"""

        def __init__(self, begin, q, chars=None):
            self.begin, self.q = begin, q
            base = d_chars if chars is None else chars
            self.chars_ptr = base.data_ptr() + (begin * K if d_offsets is None else 0)
            self.off_ptr = d_offsets.data_ptr() + 8 * begin if d_offsets is not None else 0
            self.cap = max(q // 64, 1024)  # the list's capacity
            # (a small piece may fall below the size from which batches are searched in seed order: dense results then)
            self.ordered = bool(ordered and g.search_hits_is_ordered(d_offsets is not None, K, q)) or (amino_lookup and q >= (1 << 20))
            self.form = None
            self.hits = self.listed = 0
            self.windowed = False
            self.events = []  # per recorded step: (search begin, search end, locate end)

    def search_part(p, form, ln=None):
        ln = ln or lanes[0]
        if not locate:
            g.search_hits(p.chars_ptr, p.off_ptr, K, p.q, 0, ln.counts.data_ptr(), ln.stream)
        elif form == "list":
            g.search_hits_compact(p.chars_ptr, p.off_ptr, K, p.q, ln.hit_kmers.data_ptr(), ln.hit_ranges.data_ptr(), p.cap,
                                  ln.num_hits.data_ptr(), stream=ln.stream)
        elif form == "order":
            # (round 6: with the 32-bit counts in search order beside the ranges where they are exact -- images below 2^32
            # positions --, so that the scan that follows reads 4 instead of 16 bytes per k-mer)
            g.search_hits_in_order(p.chars_ptr, p.off_ptr, K, p.q, ln.order_kmers.data_ptr(), ln.ranges.data_ptr(), stream=ln.stream,
                                   d_order_counts=ln.counts.data_ptr() if narrow_counts else 0)
        elif narrow_counts:
            # the hit offsets are scanned from the counts and the locate reads the range of a k-mer only when it has hits: the
            # ranges of the others need not be written
            g.search_hits_sparse(p.chars_ptr, p.off_ptr, K, p.q, ln.ranges.data_ptr(), ln.counts.data_ptr(), ln.stream)
        else:
            g.search_hits(p.chars_ptr, p.off_ptr, K, p.q, ln.ranges.data_ptr(), 0, ln.stream)

    list_tail = True  # the tail of a list step in one launch (awfmGpuListLocateOnDevice)

    def sorted_list(ln):
        """(k-mer numbers, ranges) of a lane's list in k-mer order"""
        return (ln.sorted_kmers, ln.sorted_ranges) if list_tail else (ln.hit_kmers, ln.hit_ranges)

    def offsets_part(p, form, ln=None, pos=None):
        """hit offsets on the device; returns (ranges, offsets, entries) the locate reads.  The list form: the whole tail -- the
        list in k-mer order, its offsets and, when `pos` is given, the positions -- in one call"""
        ln = ln or lanes[0]
        if form == "list" and list_tail:
            g.list_locate_on_device(ln.hit_kmers.data_ptr(), ln.hit_ranges.data_ptr(), p.cap, ln.num_hits.data_ptr(), p.q,
                                    ln.sorted_kmers.data_ptr(), ln.sorted_ranges.data_ptr(), ln.hit_off_c.data_ptr(),
                                    pos.numel() if pos is not None else 0, pos.data_ptr() if pos is not None else 0, ln.stream)
            return ln.sorted_ranges, ln.hit_off_c, p.cap
        if form == "list":
            g.sort_hits_on_device(ln.hit_kmers.data_ptr(), ln.hit_ranges.data_ptr(), p.cap, ln.num_hits.data_ptr(), p.q, ln.stream)
            g.hit_offsets_on_device(0, ln.hit_ranges.data_ptr(), p.cap, ln.hit_off_c.data_ptr(), ln.scratch.data_ptr(), ln.stream)
            return ln.hit_ranges, ln.hit_off_c, p.cap
        if form in ("dense", "order") and narrow_counts:
            g.hit_offsets_on_device(ln.counts.data_ptr(), 0, p.q, ln.hit_off.data_ptr(), ln.scratch.data_ptr(), ln.stream)
        else:
            g.hit_offsets_on_device(0, ln.ranges.data_ptr(), p.q, ln.hit_off.data_ptr(), ln.scratch.data_ptr(), ln.stream)
        return ln.ranges, ln.hit_off, p.q

    def dense_or_order(p):
        """a batch most of whose k-mers have hits: results in search order -- unless it is a mixed-length batch, whose dense form
        is the lookup kernel's when its sample says so (a round's ranges stored in whole lines) and whose search-order form is
        the 16-byte records': whichever one search call of each, timed here, takes less (the untimed probe chooses the form,
        the timed steps run it)"""
        if d_offsets is None:
            return "order"
        took = {}
        for f in ("dense", "order"):
            for _ in range(3):  # (the lookup prediction settles within two searches)
                search_part(p, f)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(lanes[0].torch_stream)
            search_part(p, f)
            b.record(lanes[0].torch_stream)
            torch.cuda.synchronize()
            took[f] = a.elapsed_time(b)
        p.form_probe_ms = {k: round(v, 3) for k, v in took.items()}
        return "dense" if took["dense"] <= took["order"] else "order"

    def probe(p, force=None):
        """one untimed, synchronous step: which form this piece's results take, how many hits there are, buffers to size"""
        p.form, p.windowed = "dense", False
        if not locate:
            torch.cuda.synchronize()
            search_part(p, "dense")
            torch.cuda.synchronize()
            return
        form = force or ("list" if have_list and p.ordered else "dense")
        torch.cuda.synchronize()  # (what the default stream did to the buffers is over before the lane's stream touches them)
        if form == "list":
            search_part(p, "list")
            torch.cuda.synchronize()
            p.listed = int(d_num_hits.item())
            if p.listed > p.cap:  # not a sparse batch after all
                form = dense_or_order(p) if have_order and p.ordered else "dense"
        if form != "list":
            search_part(p, form)
        ranges, offsets, entries = offsets_part(p, form)
        torch.cuda.synchronize()
        p.hits = int(offsets[entries].item())
        if form != "list":
            p.listed = 0
            if force is None and form == "dense" and have_order and p.ordered and p.hits >= p.q // 4:
                return probe(p, dense_or_order(p))  # most k-mers have hits: results in search order
        p.form = form
        if form == "list" and not os.environ.get("AWFM_BENCH_WIDE_LIST"):
            # the list's capacity for the timed steps: what the probe listed + a quarter (a caller sizes its list by what its
            # batches yield; every pass over the list -- fill, sort, scan -- is a pass over the capacity)
            p.cap = min(p.cap, max(1024, -(-(p.listed * 5 // 4) // 1024) * 1024))
        p.windowed = p.hits > WINDOW_HITS  # a hit list beyond what is kept resident: window by window (awfmGpuLocateWindow)
        ensure_positions(WINDOW_HITS if p.windowed else p.hits)
        torch.cuda.synchronize()

    def step(p, record=False, ln=None):
        ln = ln or lanes[0]
        if record:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record(ln.torch_stream)
        search_part(p, p.form, ln)
        if record:
            ev[1].record(ln.torch_stream)
        if locate and not p.windowed and p.form == "list" and list_tail:
            offsets_part(p, p.form, ln, ln.pos)
        elif locate and not p.windowed:
            ranges, offsets, entries = offsets_part(p, p.form, ln)
            g.locate_on_device(ranges.data_ptr(), offsets.data_ptr(), entries, ln.pos.numel(), ln.pos.data_ptr(), ln.stream)
        elif locate:
            # (the one form with host waits: a hit list of 5 * 10^9 positions is located in windows, whose boundaries in
            # k-mers are found from the offsets; lane 0 only)
            stream = lanes[0].stream
            total = g.hit_offsets_from_counts(d_counts.data_ptr(), p.q, d_hit_off.data_ptr(), d_scratch.data_ptr(), stream) if narrow_counts \
                else g.hit_offsets(d_ranges.data_ptr(), p.q, d_hit_off.data_ptr(), d_scratch.data_ptr(), stream)
            bounds = torch.arange(0, total + WINDOW_HITS, WINDOW_HITS, dtype=torch.int64, device=dev).clamp_(max=total)
            lanes[0].torch_stream.synchronize()
            cut = torch.searchsorted(d_hit_off[: p.q + 1], bounds, right=True).cpu().tolist()  # first k-mer whose list ends after the bound
            bounds = bounds.cpu().tolist()
            p.first_window = max(cut[1] - 1, 0)  # k-mers whose whole list lies in the first window
            for w in range(len(bounds) - 1):
                qb, qe = max(cut[w] - 1, 0), min(cut[w + 1], p.q)
                g.locate_window(d_ranges.data_ptr(), d_hit_off.data_ptr(), qb, qe, bounds[w], bounds[w + 1], pos_buf["t"].data_ptr(), stream)
                if w == 0 and getattr(p, "keep_first_window", False):
                    lanes[0].torch_stream.synchronize()
                    p.window0 = pos_buf["t"][: bounds[1]].clone()
                    torch.cuda.synchronize()
        if record:
            ev[2].record(ln.torch_stream)
            p.events.append(ev)

    def run_steps(p, count):
        """`count` steps of a piece, alternating between the lanes so that the LAST one lands in lane 0 (whose buffers every
        check reads); the caller synchronises the device before and after"""
        use = lanes if not p.windowed else lanes[:1]
        for i in range(count):
            step(p, False, use[(count - 1 - i) % len(use)])

    def check_against_probe(p, steps_run=None):
        """after a loop of steps: what the device read as the list's length and the number of hits is what the probe saw -- in
        every lane the loop used, and the lanes hold the same results"""
        if not locate or p.windowed:
            return
        used = lanes[: min(len(lanes), steps_run)] if steps_run else lanes[:1]
        for ln in used:
            if p.form == "list":
                assert int(ln.num_hits.item()) == p.listed, "the list's length changed between the probe and the timed steps"
                total = int(ln.hit_off_c[p.cap].item())
                assert ln.primary or (torch.equal(sorted_list(ln)[0][: p.listed], sorted_list(lanes[0])[0][: p.listed]) and torch.equal(ln.pos[: p.hits], pos_buf["t"][: p.hits])), \
                    "two lanes hold different results of the same batch"
            else:
                total = int(ln.hit_off[p.q].item())
                assert ln.primary or p.form == "order" or torch.equal(ln.pos[: p.hits], pos_buf["t"][: p.hits]), "two lanes hold different positions"
            assert total == p.hits and total <= ln.pos.numel(), "the number of hits changed between the probe and the timed steps"

    def to_dense(p):
        """the dense form of the piece's last step -- ranges ({1, 0} where there is no hit), counts, hit offsets under every
        k-mer number of the piece (entries [0, q) of d_ranges / d_counts / d_hit_off), positions in k-mer order -- whatever
        form the step used; outside any timed region: every check reads this"""
        q = p.q
        if not locate:
            return None
        pos = pos_buf["t"]
        if p.form == "order":
            kmers = d_order_kmers[:q].to(torch.int64)
            assert int(torch.bincount(kmers, minlength=q).max().item()) == 1, "a k-mer is missing from the order or listed twice"
            order_off = d_hit_off[: q + 1].clone()
            lens = order_off[1:] - order_off[:-1]
            counts = torch.zeros(q, dtype=torch.int64, device=dev)
            counts[kmers] = lens
            dense_off = torch.zeros(q + 1, dtype=torch.int64, device=dev)
            torch.cumsum(counts, 0, out=dense_off[1:])
            dense_ranges = torch.empty(q, 2, dtype=torch.int64, device=dev)
            dense_ranges[kmers] = d_ranges[: 2 * q].view(q, 2)
            dense_pos = torch.empty(max(p.hits, 1), dtype=torch.int64, device=dev)
            step_q = 1 << 24
            for b in range(0, q, step_q):  # destination of every hit: its k-mer's dense offset + its rank in the list
                e = min(q, b + step_q)
                lo, hi = int(order_off[b].item()), int(order_off[e].item())
                if hi > lo:
                    shift = torch.repeat_interleave(dense_off[:-1][kmers[b:e]] - order_off[b:e], lens[b:e])
                    dense_pos[shift + torch.arange(lo, hi, dtype=torch.int64, device=dev)] = pos[lo:hi]
            d_ranges[: 2 * q].copy_(dense_ranges.view(-1))
            d_counts[:q].copy_(counts.to(torch.int32))
            d_hit_off[: q + 1].copy_(dense_off)
            return dense_pos
        if p.form == "list":
            m = p.listed
            list_kmers, list_ranges = sorted_list(lanes[0])
            kmers = list_kmers[:m].to(torch.int64)
            assert m == 0 or bool((kmers[1:] > kmers[:-1]).all()), "the hit list is not in k-mer order"
            assert m == p.cap or int(list_kmers[m].item()) == -1, "an entry behind the list's length"
            lens = d_hit_off_c[1:m + 1] - d_hit_off_c[:m]
            d_counts[:q].zero_()
            d_counts[:q][kmers] = lens.to(torch.int32)
            d_hit_off[0] = 0
            torch.cumsum(d_counts[:q].to(torch.int64), 0, out=d_hit_off[1:q + 1])
            dense = d_ranges[: 2 * q].view(q, 2)
            dense[:, 0] = 1
            dense[:, 1] = 0
            dense[kmers] = list_ranges.view(-1, 2)[:m]
            assert int(d_hit_off[q].item()) == p.hits
            return pos
        # dense: ranges were written for the k-mers with hits only (awfmGpuSearchHitsSparse): "no hit" for the rest
        lens = d_hit_off[1:q + 1] - d_hit_off[:q]
        if narrow_counts:
            d_ranges[: 2 * q].view(q, 2)[lens == 0] = torch.tensor([1, 0], dtype=torch.int64, device=dev)
        else:
            d_counts[:q].copy_(lens.clamp(max=0xFFFFFFFF).to(torch.int32))
        return pos

    def barrier():
        shard.barrier(world, torch.cuda.synchronize)

    whole = Piece(0, Q)
    probe(whole)
    if args.streams > 1 and not whole.windowed:
        lanes.extend(Lane(False) for _ in range(args.streams - 1))
        ensure_positions(whole.hits)
    # the warm-up steps carry the events that split a step into its search call and its locate kernels (reporting); the timed
    # steps carry none: a recorded event is a packet of its own in the queue, about 5 us of idle device each
    for _ in range(max(args.warmup, 1)):
        step(whole, True)
    torch.cuda.synchronize()
    g.ordered_kernel_log()  # the log of the library's own kernel brackets starts with the timed steps
    barrier()
    t_start = time.perf_counter()
    run_steps(whole, args.steps)
    barrier()
    elapsed = time.perf_counter() - t_start
    elapsed = shard.max_over_ranks(elapsed, world, dev)
    ms_per_step = elapsed * 1e3 / args.steps
    value = batch_total / (elapsed / args.steps) / 1e6  # Mkmers/s over all ranks
    check_against_probe(whole, args.steps)
    search_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in whole.events]))
    locate_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in whole.events])) if locate else 0.0
    kernel_log = g.ordered_kernel_log() if ordered else []
    assert not ordered or len(kernel_log) == min(args.steps, 1024), "the library logged another number of searches than were timed"
    lookup_first = bool(ordered and g.last_ordered_kernel_is_lookup())  # the dominant kernel of every step was encodeLookupKernel
    # (no seed-order path: the hits-only search is awfmGpuSearch, which takes large batches through exactLookupSearchKernel)
    exact_looked_up = bool(not ordered and g.last_search_was_exact_lookup())
    # which front end(s) the last timed step launched: 0 both (the sample's verdict stays on the device), 1 the lookup kernel
    # only / 2 the ordered kernels only (the verdict of an earlier step had reached the host: awfmGpuLastLookupFront), -1: no sample
    lookup_front = g.last_lookup_front()
    # (a mixed-length batch below the seed-order size on an image with its tables per k-mer length: the lookup kernel alone)
    small_mixed_lookup = bool(d_offsets is not None and not ordered and not amino and g.length_tables[0] and g.last_ordered_kernel_is_lookup())
    lookup_kept = g.last_ordered_kept() if lookup_first else 0  # before any other search re-uses the scratch
    amino_looked_up = bool(amino_lookup and whole.ordered and g.last_ordered_kernel_is_lookup())  # aminoLookupSearchKernel did the timed steps (its sample said so)
    lookup_kept_amino = g.last_ordered_kept() if amino_looked_up else 0  # k-mers still alive after their table entry
    ordered_ms = [(f if lookup_first else k) for f, k in kernel_log]
    after_lookup_ms = float(np.mean([k for _, k in kernel_log])) if lookup_first else None
    state = {"hits": whole.hits, "listed": whole.form == "list", "in_order": whole.form == "order", "windowed": whole.windowed,
             "form": whole.form}
    if whole.windowed:  # the positions of the first window, kept for the oracle check of the k-mers that lie in it
        whole.keep_first_window = True
        step(whole)
        torch.cuda.synchronize()
        whole.keep_first_window = False
        state["window0"], state["first_window"] = whole.window0, whole.first_window
    # every check below (oracle sample, digests, pipelines, dumps) reads the dense form, made here outside the timed region
    state["positions"] = to_dense(whole)
    state["sparse"] = False  # (the dense arrays are complete: ranges of the k-mers without hits are {1, 0}, counts are written)

    if args.dump_dir:  # testing: this rank's shard results, for a check against the oracle outside the bench
        os.makedirs(args.dump_dir, exist_ok=True)
        dump = {"first": np.uint64(first), "queries": np.uint64(Q)}
        if args.mode == "count":
            dump["counts"] = d_counts.cpu().numpy().view(np.uint32)
        else:
            dump["ranges"] = d_ranges.cpu().numpy().view(np.uint64).reshape(Q, 2)
            dump["hit_offsets"] = d_hit_off.cpu().numpy().view(np.uint64)
            if not state["windowed"]:
                dump["positions"] = state["positions"][: state["hits"]].cpu().numpy().view(np.uint64)
        np.savez(os.path.join(args.dump_dir, f"rank{rank}.npz"), **dump)

    # ---- digests of this rank's results; rank 0 checks all ranks' against the committed digests of the 1-rank run ----
    kmer_name = f"{args.mixed_lengths[0]}-{args.mixed_lengths[1]}" if args.workload == "mixed" else str(K)

    def describe(f, c):
        return digest.key(args.alphabet + ("" if args.text == "uniform" else "-" + args.text), args.workload, args.mode, n,
                          kmer_name, args.seed_k, args.sa_ratio, f, c)

    if args.mode == "count":
        mine = (first, Q, digest.counts_digest(first, d_counts), None)
    else:
        shard_counts = (d_hit_off[1:] - d_hit_off[:-1])
        mine = (first, Q, digest.counts_digest(first, shard_counts),
                None if state["windowed"] else digest.positions_digest(first, d_hit_off, state["positions"][: max(state["hits"], 1)]))
        del shard_counts
    all_digests = shard.gather_objects(mine, world)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    digest_check = digest.check_against_golden(all_digests, digest.load_golden(), describe)  # raises on a mismatch
    digest_check["per_rank"] = [{"first": f, "count": c, "counts": f"{dc:016x}", "positions": f"{dp:016x}" if dp is not None else None}
                                for f, c, dc, dp in all_digests]
    if args.record_digests:
        known = json.load(open(args.record_digests)) if os.path.exists(args.record_digests) else {}
        for f, c, dc, dp in all_digests:
            known[describe(f, c)] = {"counts": f"{dc:016x}", "positions": f"{dp:016x}" if dp is not None else None}
        os.makedirs(os.path.dirname(os.path.abspath(args.record_digests)), exist_ok=True)
        json.dump(known, open(args.record_digests, "w"), indent=1, sort_keys=True)

    kdesc = f"{args.mixed_lengths[0]}..{args.mixed_lengths[1]}-mers" if args.workload == "mixed" else f"{K}-mers"
    # ---- roofline of the dominant kernel ----
    tally = g.search_tally(d_chars.data_ptr(), off_ptr, K, Q)
    rank_bytes = RANK_BYTES_AMINO if amino else RANK_BYTES_DNA
    alg_bytes = tally["chars"] + 16 * tally["seeded"] + rank_bytes * tally["blocks"] + 16 * Q
    per_query = {"steps": round(tally["steps"] / Q, 4), "distinct_blocks": round(tally["blocks"] / Q, 4),
                 "seeded": round(tally["seeded"] / Q, 4), "bytes": round(alg_bytes / Q, 1)}
    upper_bytes = tally["chars"] + 16 * tally["seeded"] + rank_bytes * 2 * tally["steps"] + 16 * Q
    # Counter-derived figures come from rocprofv3 PMC passes of THIS command line (scripts/profile_bench.sh ->
    # scripts/collect_profiles.py -> profiles/<round>/); a bench run cannot collect them itself, so they are attached only
    # when the arguments are the profiled ones, and labelled with their source.
    prof_name = None
    if (args.device_seed_k < 0 and args.device_dense_sa is None and not amino and n == 3_100_000_000 and Q == 100_000_000
            and K == 21 and args.seed_k == 12 and args.sa_ratio == 8 and args.mode == "locate" and args.text == "uniform"):
        prof_name = {"random": "default", "planted": "planted"}.get(args.workload)
    if (args.device_seed_k < 0 and args.device_dense_sa is None and not amino and n == 3_100_000_000 and Q == 100_000_000
            and K == 21 and args.seed_k == 12 and args.sa_ratio == 8 and args.mode == "count" and args.text == "uniform" and args.workload == "planted"):
        prof_name = "planted_count"
    if (args.device_seed_k < 0 and args.device_dense_sa is None and not amino and n == 6_200_000_000 and Q == 100_000_000
            and K == 21 and args.seed_k == 12 and args.sa_ratio == 8 and args.mode == "locate" and args.text == "uniform"):
        prof_name = {"random": "wide", "planted": "wide_planted"}.get(args.workload)
    if (args.device_seed_k < 0 and args.device_dense_sa is None and not amino and n == 3_100_000_000 and Q == 100_000_000
            and args.workload == "mixed" and args.seed_k == 12 and args.sa_ratio == 8 and args.mode == "count" and args.text == "uniform"):
        prof_name = "mixed"
    if (args.device_seed_k < 0 and args.device_dense_sa is None and not amino and n == 3_100_000_000 and Q == 100_000_000
            and K == 21 and args.seed_k == 12 and args.sa_ratio == 8 and args.text == "repetitive"):
        prof_name = {("unique", "locate"): "repetitive_unique", ("planted", "count"): "repetitive_planted"}.get((args.workload, args.mode))
    if any(k.startswith("AWFM_GPU_") and k not in ("AWFM_GPU_TIME_ORDERED", "AWFM_GPU_DEVICE") for k in os.environ):
        prof_name = None  # a measurement knob is set: the profiled run was of the default code path
    not_this_run = "rocprofv3 PMC passes of this command line on another run of the same code (scripts/profile_bench.sh), not measured by this run"
    if ordered:
        # The search of a large batch is several launches (k-mer encoding + bucket count, partition, orderedSearchKernel).
        # The dominant kernel starts from a deeper table and serves repeated block reads out of the L2, so the reference
        # algorithm's bytes are not what it has to move: its roofline is its COMPULSORY traffic -- every 128-B line once per search level that needs
        # it, the sorted records, the results -- over its own HIP-event time.
        mixed_lookup = lookup_first and d_offsets is not None  # mixedLookupSearchKernel: its own tally (below)
        lines = g.search_hits_line_tally(d_chars.data_ptr(), off_ptr, K, Q) if not mixed_lookup else \
            g.mixed_lookup_line_tally(d_chars.data_ptr(), off_ptr, Q)
        if mixed_lookup:
            lines.update(seed_table_lines=0, ordered_kmers=Q - lines["general_kmers"], record_bytes_per_kmer=0)
        # what the search stores per result, by the form the timed steps used
        stored = {"list": 20 * lines["kmers_with_hits"], "order": 20 * lines["ordered_kmers"],
                  "dense": (16 + (4 if narrow_counts else 0)) * lines["kmers_with_hits"]}[whole.form] if locate else 4 * lines["kmers_with_hits"]
        compulsory = (128 * (lines["seed_table_lines"] + lines["deep_table_lines"] + lines["pair_level_lines"] + lines["nuc_level_lines"])
                      + lines["record_bytes_per_kmer"] * lines["ordered_kmers"] + stored)
        dom_ms = float(np.mean(ordered_ms))
        dom_name = "orderedSearchKernel"
        what = ("distinct (search level, 128-B line) pairs the kernel reads, tallied on the device by an "
                "instrumented launch of the same kernel on the same sorted batch (awfmGpuSearchHitsLineTally), "
                "x 128 B, + sorted records and keys read + results stored")
        if mixed_lookup:
            # A mixed-length batch that took "lookup first" (DESIGN.md 4c): ONE table entry per k-mer -- from the table of its
            # own length (k-mers shorter than the deeper table's) or from the deeper table --, then the steps of the k-mers
            # still alive.  Its compulsory bytes: characters + offsets + every distinct line of those tables once + every
            # distinct (level, line) of the survivors' block reads + the results; tallied by an instrumented pass over the
            # same k-mers (awfmGpuMixedLookupLineTally).
            dom_name = "mixedLookupSearchKernel"
            lines["characters"] = total_chars
            streamed = total_chars + 8 * (Q + 1)
            compulsory = streamed + 128 * (lines["length_table_lines"] + lines["deep_table_lines"] + lines["pair_level_lines"]
                                           + lines["nuc_level_lines"]) + stored
            lines = dict(lines, kmers_kept=int(lookup_kept), characters_read=int(lines["characters"]), offsets_read_bytes=8 * (Q + 1))
            what = ("the k-mers' characters and offsets + 128 B x (the distinct lines of the length tables and of the deeper table the "
                    "batch's k-mers need + the distinct (search level, line) pairs of the block reads of the k-mers still alive after "
                    "their entry), tallied on the device by awfmGpuMixedLookupLineTally, + the results stored")
        elif lookup_first:
            # The batch was one for "lookup first"  and the kernel that looks the table entries up also searches
            # the few k-mers that are still alive after them (lookupSearchKernel).  Its compulsory bytes: the characters +
            # every distinct table line once (the tally's deep_table_lines: the same entries, whatever the order) + every
            # distinct (level, line) of the block reads of the k-mers kept (the tally's pair / one-letter level lines: the
            # same steps, whatever the order) + the results.
            dom_name = "lookupSearchKernel"
            compulsory = Q * K + 128 * (lines["deep_table_lines"] + lines["pair_level_lines"] + lines["nuc_level_lines"]) + stored
            lines = dict(lines, kmers_kept=int(lookup_kept), characters_read=int(Q * K))
            what = ("the k-mers' characters + 128 B x (the distinct lines of the deeper table the batch's k-mers need + the distinct "
                    "(search level, line) pairs of the block reads of the k-mers still alive after the table), tallied on the "
                    "device by awfmGpuSearchHitsLineTally, + the results stored")
        achieved = compulsory / (dom_ms * 1e-3) / 1e9
        # NEEDED bytes: what the kernel consumes -- a table lookup is an 8-byte entry (16 from 2^32 positions), not the 128-B
        # line it arrives in; the block reads of the search levels stay at their distinct lines.  traffic / needed says how
        # much of what the kernel moves is the rest of a line nobody asked for.
        entry_bytes = 8 if ix.bwt_length < (1 << 36) else 16  # (round 6: the packed entries of images of 2^32 .. 2^36 positions are 8 bytes too)
        if mixed_lookup:
            needed = streamed + 8 * lines["ordered_kmers"] + 128 * (lines["pair_level_lines"] + lines["nuc_level_lines"]) + stored
        elif lookup_first:
            needed = Q * K + entry_bytes * Q + 128 * (lines["pair_level_lines"] + lines["nuc_level_lines"]) + stored
        else:
            needed = (entry_bytes * lines["ordered_kmers"] + 128 * (lines["pair_level_lines"] + lines["nuc_level_lines"])
                      + lines["record_bytes_per_kmer"] * lines["ordered_kmers"] + stored)
        roofline = {
            "bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel_ms": round(dom_ms, 3),
            "basis": "compulsory_lines",
            "basis_note": "achieved = compulsory_bytes / kernel_ms: every distinct 128-B line the kernel needs, once (the builder's "
                          "accounting, not SURVEY 8(d)'s); the reference algorithm's bytes over THIS kernel's time would be "
                          "algorithmic_frac_of_this_kernel (> 1: the deeper table answers most k-mers in one gather, those bytes are "
                          "never read); the kernel that does execute the reference algorithm's steps is priced in reference_algorithm",
            "compulsory_bytes": int(compulsory), "needed_bytes": int(needed), "frac_needed": round(needed / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "algorithmic_frac_of_this_kernel": round(alg_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
            "compulsory": {"what": what, **lines, "result_bytes_stored": int(stored)},
            "limiter": "HBM, as a gather of random 128-B lines (the guide measures 5.5-5.8 TB/s for such reads, 0.69-0.73 of the "
                       "peak): most lines are entries of the deeper seed table, one per k-mer; see hbm_frac_measured for this "
                       "kernel's measured traffic and `l2` for the requests it sends to the L2s",
        }
        if mixed_lookup:
            # what the kernel reads AS EXECUTED: it takes the k-mers in batch order, so no two k-mers share a line the way an
            # ideal cache (the compulsory figure) would have them -- one 128-B line per table entry, one per block read of a
            # survivor's step -- a gather of random lines, whose ceiling on this chip is 0.69-0.73 of the HBM peak
            executed = streamed + 128 * (lines["ordered_kmers"] + lines["block_reads_executed"]) + stored
            roofline["executed_read_bytes"] = int(executed)
            roofline["frac_executed_reads"] = round(executed / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        counters, csrc = profile_file("counters", prof_name) if prof_name else (None, None)
        traffic, tsrc = profile_file("traffic", prof_name) if prof_name else (None, None)
        if counters and not str(counters.get("kernel", "")).startswith(dom_name):
            counters = None  # the committed counters are another kernel's (a profile from before this path existed)
        if counters and "hbm_read_bytes" in counters:
            roofline["traffic"] = int(counters["hbm_read_bytes"] + counters.get("hbm_write_bytes", 0.0))
            roofline["traffic_source"] = f"{csrc}: {not_this_run}; reads = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM), writes = WRITE_SIZE"
            roofline["traffic_over_compulsory"] = round(roofline["traffic"] / compulsory, 3)
            roofline["traffic_over_needed"] = round(roofline["traffic"] / needed, 3)
            profiled_ms = counters.get("avg_ns_kernel_trace", 0.0) / 1e6
            if profiled_ms and abs(profiled_ms - dom_ms) <= 0.15 * dom_ms:
                roofline["hbm_frac_measured"] = round(roofline["traffic"] / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            else:  # bytes of launches that ran at another speed say nothing about this run's rate
                roofline["traffic_note"] = (f"the profiled launches of this kernel averaged {profiled_ms:.2f} ms under rocprofv3 against "
                                            f"{dom_ms:.2f} ms here: no measured-traffic fraction is derived")
        if counters and "TCC_REQ_sum" in counters.get("raw", {}):
            req = counters["raw"]["TCC_REQ_sum"]
            l2_gbs = req * 128 / (dom_ms * 1e-3) / 1e9
            roofline["l2"] = {"requests": int(req), "requests_per_kmer": round(req / Q, 3), "bytes": int(req * 128),
                              "achieved": round(l2_gbs, 1), "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s",
                              "frac": round(l2_gbs / L2_GATHER_PEAK_GBS, 4),
                              "source": f"{csrc}: TCC_REQ_sum x 128 B over this run's kernel time; peak = the guide's chip-wide rate "
                                        "for rows gathered out of the XCDs' L2s (16.8-18.8 TB/s)"}
        dominant = {"name": dom_name, "ms": round(dom_ms, 3)}
        if counters:
            for key in ("l2_hit_rate", "valu_issue_frac", "wave_wait_frac", "clock_ghz_under_profiler"):
                if key in counters:
                    dominant[key] = round(counters[key], 4)
            dominant["counters_source"] = (f"{csrc} ({not_this_run}); valu_issue_frac = SQ_INSTS_VALU x 2 cycles / "
                                           "(1024 SIMDs x GRBM_GUI_ACTIVE / 8), the guide's wave64 issue cost")
        roofline["dominant_kernel"] = dominant
        # the whole call, priced by what the REFERENCE algorithm would move for this batch: a throughput figure in bytes,
        # not a roofline fraction (five sixths of those bytes never leave the L2)
        roofline["call"] = {
            "kernels": ("awfmGpuSearchHits*: fill / memset + mixedSampleAliveKernel + mixedLookupSearchKernel + the general kernel over "
                        "what it left (+ the kernels of the 16-byte-record path, which return at once)" if mixed_lookup else
                        "awfmGpuSearchHits*: lookupPrepKernel + lookupSearchKernel (+ the kernels of the other front end while no "
                        "prediction holds, which return at once) + the general kernel over what it left" if lookup_first else
                        "awfmGpuSearchHits*: fill / memset + encodeCodes4Kernel (count) + bucketScanSharesKernel + partitionKernel + "
                        "orderedSearchKernel (fixed lengths with 8-byte records; 16-byte records: encodeRecordsKernel + "
                        "partitionRecordsKernel)"),
            "ms": round(search_ms, 3), "algorithmic_bytes_per_launch": int(alg_bytes), "per_query": per_query,
            "unordered_equivalent_GBs": round(alg_bytes / (search_ms * 1e-3) / 1e9, 1),
            "unordered_equivalent_upper_bound_variant_GBs": round(upper_bytes / (search_ms * 1e-3) / 1e9, 1),
            "traffic": traffic["hbm_bytes_per_launch"] if traffic else None, "traffic_source": f"{tsrc}: {not_this_run}" if traffic else None,
        }
        if traffic:
            roofline["call"]["hbm_frac_measured"] = round(traffic["hbm_bytes_per_launch"] / (search_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    elif g.deep_seed_k:
        # The kernel that starts from the device-only deeper table -- aminoLookupSearchKernel for large fixed-length amino
        # batches, the general kernel otherwise (amino: DESIGN.md 4b): most k-mers end at their table entry,
        # so the reference algorithm's bytes are not what the kernel reads.  `frac` prices what it EXECUTES -- an instrumented
        # launch of the same kernel with the table on: a 128-B line per table lookup, 168 B per distinct block of the steps
        # behind it -- and `reference_algorithm` is the same kernel without the table (the reference's steps),
        # timed on the same batch, whose ranges must equal the timed steps'.
        _diag_before = os.environ.get("AWFM_GPU_DIAG")
        os.environ["AWFM_GPU_DIAG"] = "tally_with_deep=1"  # the library's variable of test and diagnostics hooks
        executed = g.search_tally(d_chars.data_ptr(), off_ptr, K, Q)
        os.environ.pop("AWFM_GPU_DIAG")
        if _diag_before is not None:
            os.environ["AWFM_GPU_DIAG"] = _diag_before
        deep_lookups = Q - executed["seeded"] if K >= g.deep_seed_k else 0  # fixed-length k-mers without ambiguity letters start at the deeper table
        exec_bytes = executed["chars"] + 128 * deep_lookups + 16 * executed["seeded"] + rank_bytes * executed["blocks"] + 16 * Q
        if amino_looked_up:
            # (round 5) the lookup kernel drops a k-mer at its entry when the entry's next-letter bit is clear, so it executes fewer
            # steps than the general kernel the tally instruments: what it reads AT LEAST -- the characters, a line per table entry,
            # one block per k-mer still alive after its entry, the results it stores
            exec_bytes = executed["chars"] + 128 * deep_lookups + rank_bytes * lookup_kept_amino + (20 * whole.listed if whole.form == "list" else 16 * Q)
        achieved = exec_bytes / (search_ms * 1e-3) / 1e9
        had_deep = g.deep_seed_k
        plain_ms = None
        if args.general_steps > 0:
            d_exact = torch.empty(Q * 2, dtype=torch.int64, device=dev)
            g.search(d_chars.data_ptr(), off_ptr, K, Q, d_exact.data_ptr(), 0, stream)
            g.set_deep_seed(0)
            d_plain = torch.empty(Q * 2, dtype=torch.int64, device=dev)
            g.search(d_chars.data_ptr(), off_ptr, K, Q, d_plain.data_ptr(), 0, stream)
            torch.cuda.synchronize()
            assert torch.equal(d_exact, d_plain), "the deeper table changes a range"
            events = []
            for _ in range(args.general_steps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.search_hits(d_chars.data_ptr(), off_ptr, K, Q, d_plain.data_ptr(), d_counts.data_ptr(), stream)
                e1.record()
                events.append((e0, e1))
            torch.cuda.synchronize()
            plain_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
            del d_exact, d_plain
            g.set_deep_seed(had_deep)
            torch.cuda.synchronize()
        plain_gbs = alg_bytes / (plain_ms * 1e-3) / 1e9 if plain_ms else None
        roofline = {
            "bound": "hbm", "kernel": ("aminoLookupSearchKernel" if amino_looked_up else
                                       "mixedLookupSearchKernel (priced by the general kernel's reads of the same batch: an upper bound)"
                                       if small_mixed_lookup else
                                       "exactLookupSearchKernel" if exact_looked_up else "searchKernel") + f" (device-only table of depth {had_deep})",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "kernel_ms": round(search_ms, 3), "basis": "executed_reads_lower_bound" if amino_looked_up else "executed_reads",
            "basis_note": "achieved = (characters + 128 B per lookup in the deeper table + 16 B per seed-table entry + 168 B per "
                          "distinct block of the steps executed behind the table + 16 B out) / kernel_ms, tallied by an instrumented "
                          "launch of the same kernel; the reference algorithm's bytes over this kernel's time would be "
                          "algorithmic_frac_of_this_kernel; the kernel that executes the reference's steps is in reference_algorithm",
            "executed_bytes": int(exec_bytes), "executed_per_query": {"steps": round(executed["steps"] / Q, 4), "distinct_blocks": round(executed["blocks"] / Q, 4),
                                                                      "deep_table_lookups": round(deep_lookups / Q, 4)},
            "algorithmic_bytes_per_launch": int(alg_bytes), "per_query": per_query,
            "algorithmic_frac_of_this_kernel": round(alg_bytes / (search_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
        }
        if plain_ms:
            roofline["reference_algorithm"] = {"kernel": "searchKernel (no deeper table)", "bytes": int(alg_bytes), "kernel_ms": round(plain_ms, 3),
                                               "frac": round(plain_gbs / HBM_PEAK_GBS, 4), "steps": args.general_steps,
                                               "checked": "every range equals the timed kernel's"}
            roofline.update(reference_algorithm_bytes=int(alg_bytes), reference_algorithm_kernel_ms=round(plain_ms, 3),
                            reference_algorithm_frac=round(plain_gbs / HBM_PEAK_GBS, 4))
    else:
        achieved = alg_bytes / (search_ms * 1e-3) / 1e9
        roofline = {
            "bound": "hbm", "kernel": "searchKernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel_ms": round(search_ms, 3),
            "algorithmic_bytes_per_launch": int(alg_bytes), "per_query": per_query,
            "upper_bound_variant_GBs": round(upper_bytes / (search_ms * 1e-3) / 1e9, 1),
        }
    if not ordered:
        gname = None
        if amino and Q == 50_000_000 and K == 10 and args.seed_k == 5 and n in (200_000_000, 2_000_000_000):
            gname = "amino" if n == 200_000_000 else "amino_2e9"
        if gname and not any(k.startswith("AWFM_GPU_") and k != "AWFM_GPU_DEVICE" for k in os.environ):
            traffic, tsrc = profile_file("traffic", gname)
            if traffic:
                roofline["traffic"] = traffic["hbm_bytes_per_launch"]
                roofline["traffic_source"] = f"{tsrc}: {not_this_run}"
                roofline["hbm_frac_measured"] = round(roofline["traffic"] / (search_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)

    # ---- roofline_general: the exact-range general kernel on the same batch (awfmGpuSearch: every k-mer's final range,
    # the reference's own stopping rule) -- one random 128-B line per block read, HBM-bound, priced by the algorithmic
    # bytes.  Its ranges of the k-mers with hits must be the ones the timed steps produced. ----
    roofline_general = None
    if ordered and args.general_steps > 0:
        # priced by the reference algorithm's bytes, so it must execute the reference algorithm's steps: the device-only
        # deeper table (which answers the first deepK - seedK steps from one entry) is dropped for this measurement
        had_deep = g.deep_seed_k
        if had_deep:
            g.set_deep_seed(0)
        d_exact = torch.empty(Q * 2, dtype=torch.int64, device=dev)
        g.search(d_chars.data_ptr(), off_ptr, K, Q, d_exact.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        gen_state_before = gpu_state()
        events = []
        for _ in range(args.general_steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.search(d_chars.data_ptr(), off_ptr, K, Q, d_exact.data_ptr(), 0, stream)
            e1.record()
            events.append((e0, e1))
        time.sleep(0.03)  # (the launches are in flight: what the device reports of itself now is what it runs them at)
        gen_state_during = gpu_state()
        torch.cuda.synchronize()
        gen_each = [a.elapsed_time(b) for a, b in events]
        gen_ms = float(np.mean(gen_each))
        gen_state_after = gpu_state()
        if args.mode == "locate":
            ex = d_exact.view(Q, 2)
            has = ex[:, 0] <= ex[:, 1]
            assert int(has.sum().item()) == int((d_hit_off[1:] > d_hit_off[:-1]).sum().item()), "the exact kernel finds other k-mers"
            assert torch.equal(ex[has], d_ranges.view(Q, 2)[has]), "exact ranges differ from the seed-order path's"
            del ex, has
        gen_gbs = alg_bytes / (gen_ms * 1e-3) / 1e9
        roofline_general = {
            "bound": "hbm", "kernel": "searchKernel (awfmGpuSearch: exact final range of every k-mer, pair steps)",
            "achieved": round(gen_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gen_gbs / HBM_PEAK_GBS, 4),
            "traffic": None, "kernel_ms": round(gen_ms, 3), "steps": args.general_steps,
            "algorithmic_bytes_per_launch": int(alg_bytes), "per_query": per_query,
            "Mkmers_per_s": round(Q / gen_ms / 1e3, 1), "seed_table_k": args.seed_k,
            "checked": "ranges of every k-mer with hits equal the timed steps'" if args.mode == "locate" else None,
            # round 5's verdict: this kernel's time moved by 11-14 % between boxes and nobody had logged what the box was doing.
            # Every launch's own time, and the clocks / partition modes the device reported right before and right after them
            "kernel_ms_each": [round(x, 3) for x in gen_each],
            "kernel_ms_min_median_max": [round(min(gen_each), 3), round(float(np.median(gen_each)), 3), round(max(gen_each), 3)],
            "spread": round((max(gen_each) - min(gen_each)) / float(np.median(gen_each)), 4),
            "gpu_state_before": gen_state_before, "gpu_state_during": gen_state_during, "gpu_state_after": gen_state_after,
        }
        if prof_name == "default":
            traffic, tsrc = profile_file("traffic", "general_pair")
            if traffic:
                roofline_general["traffic"] = traffic["hbm_bytes_per_launch"]
                roofline_general["traffic_source"] = f"{tsrc}: {not_this_run} (AWFM_GPU_ORDERED=0 --mode count)"
                roofline_general["hbm_frac_measured"] = round(traffic["hbm_bytes_per_launch"] / (gen_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if had_deep:
            g.set_deep_seed(had_deep)
            torch.cuda.synchronize()
            # ... and the same call -- awfmGpuSearch: the exact final range of EVERY k-mer, the first empty range of the reference's
            # stepping for the k-mers without hits -- with the device-only tables back in place: large batches take
            # exactLookupSearchKernel (one table entry per k-mer, exact pair steps behind it); every range must equal the
            # general kernel's
            d_tables = torch.empty(Q * 2, dtype=torch.int64, device=dev)
            g.search(d_chars.data_ptr(), off_ptr, K, Q, d_tables.data_ptr(), 0, stream)
            torch.cuda.synchronize()
            events = []
            for _ in range(args.general_steps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.search(d_chars.data_ptr(), off_ptr, K, Q, d_tables.data_ptr(), 0, stream)
                e1.record()
                events.append((e0, e1))
            torch.cuda.synchronize()
            tables_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
            assert torch.equal(d_tables, d_exact), "awfmGpuSearch through the tables differs from the general kernel in some k-mer's final range"
            roofline_general["through_the_tables"] = {
                "kernel": "exactLookupSearchKernel (awfmGpuSearch with the device-only tables: one entry per k-mer, exact pair steps behind it)",
                "kernel_ms": round(tables_ms, 3), "Mkmers_per_s": round(Q / tables_ms / 1e3, 1), "steps": args.general_steps,
                "checked": "the final range of every k-mer of the batch equals the general kernel's (no deeper table)"}
            del d_tables
        del d_exact
        # the same figures inside `roofline`, flat, for readers that keep its scalars only: SURVEY 8(d)'s bytes over the time of
        # the kernel that reads them
        roofline["reference_algorithm"] = {"kernel": "searchKernel", "bytes": int(alg_bytes), "kernel_ms": round(gen_ms, 3),
                                           "frac": round(gen_gbs / HBM_PEAK_GBS, 4), "steps": args.general_steps}
        roofline["reference_algorithm_kernel"] = "searchKernel (awfmGpuSearch, exact ranges, no deeper table)"
        roofline["reference_algorithm_bytes"] = int(alg_bytes)
        roofline["reference_algorithm_kernel_ms"] = round(gen_ms, 3)
        roofline["reference_algorithm_frac"] = round(gen_gbs / HBM_PEAK_GBS, 4)
        if "through_the_tables" in roofline_general:
            roofline["exact_ranges_through_the_tables_ms"] = roofline_general["through_the_tables"]["kernel_ms"]
            roofline["exact_ranges_through_the_tables_Mkmers_per_s"] = roofline_general["through_the_tables"]["Mkmers_per_s"]

    # ---- CPU baseline: the oracle on this box's host cores, bounded sample, parity-gated ----
    cpu = None
    if not args.no_cpu:
        from oracle import oracle as O
        oalpha = O.AMINO if amino else O.DNA
        oi = O.Index.wrap(oalpha, args.sa_ratio, args.seed_k, ix.bwt_length, ix.blocks(), ix.prefix_sums(),
                          ix.seed_table(), ix.packed_sa())
        # host threads: the box may grant fewer CPUs than it shows (cgroup quota); the path is DRAM-latency
        # bound, so a few threads per granted CPU are tried on a small sample and the fastest is kept
        granted = os.cpu_count() or 1
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                granted = max(1, min(granted, -(-int(quota) // int(period))))
        except (OSError, ValueError):
            pass
        candidates = sorted({min(os.cpu_count() or 1, granted * f) for f in (1, 2, 4)})
        cores = candidates[0]

        def run_sample(m, threads=None):
            threads = threads or cores
            if d_offsets is not None:
                offsets = d_offsets[: m + 1].cpu().numpy().view(np.uint64)
                chars = d_chars[: int(offsets[-1])].cpu().numpy()
            else:
                chars = d_chars[: m * K].cpu().numpy()
                offsets = np.arange(m + 1, dtype=np.uint64) * np.uint64(K)
            t0 = time.perf_counter()
            sp, ep, cnt, tl = oi.batch_search(chars, offsets, threads=threads)
            if args.mode == "locate":
                ho, pos, tl2 = oi.batch_locate(sp, ep, threads=threads)
            dt = time.perf_counter() - t0
            # parity gate on the sample: ranges and (for locate) hit positions in BWT order
            if args.mode == "count":  # counting asks for the counts only
                assert np.array_equal(d_counts[:m].cpu().numpy().view(np.uint32), cnt), "GPU counts differ from the oracle"
            else:
                gr = d_ranges[: 2 * m].cpu().numpy().view(np.uint64).reshape(m, 2)
                hit = cnt > 0  # hits-only contract: exact ranges for queries with hits ...
                assert np.array_equal(gr[hit, 0], sp[hit]) and np.array_equal(gr[hit, 1], ep[hit]), "GPU ranges differ from the oracle"
                # ... and count 0 and some empty range for the others
                assert np.array_equal(d_counts[:m].cpu().numpy().view(np.uint32), cnt), "GPU counts differ from the oracle"
                assert np.all(gr[~hit, 0] > gr[~hit, 1]), "GPU reports hits the oracle does not have"
            if args.mode == "locate":
                gho = d_hit_off[: m + 1].cpu().numpy().view(np.uint64)
                assert np.array_equal(gho, ho), "GPU hit offsets differ from the oracle"
                gp = (state["window0"] if state["windowed"] else state["positions"])[: int(ho[-1])].cpu().numpy().view(np.uint64)
                assert np.array_equal(gp, pos), "GPU positions differ from the oracle"
            return dt, tl

        # grow the sample until it costs about --cpu-seconds of wall time (thread start-up and first-touch
        # page faults dominate tiny samples)
        sample_cap = min(Q, 50_000_000)
        if state["windowed"]:  # positions are checked for the k-mers whose lists lie in the first window
            sample_cap = max(1, min(sample_cap, state["first_window"]))
        m = min(sample_cap, 2_000_000)
        best = None
        for c in candidates:  # thread-count probe: which oversubscription, if any, hides the DRAM latency best
            dt, tl = run_sample(m, c)
            if best is None or dt < best[0]:
                best = (dt, c)
        over = best[1]
        # the headline baseline runs one thread per GRANTED CPU (round 5's verdict: `cores` must be cores; the oversubscribed
        # figure, which is the faster one on a latency-bound path, is reported beside it)
        cores = granted
        dt, tl = run_sample(m)
        for _ in range(4):
            if dt >= args.cpu_seconds / 2 or m >= sample_cap:
                break
            m = int(min(sample_cap, max(2 * m, m * args.cpu_seconds / max(dt, 1e-3))))
            dt, tl = run_sample(m)
        # the port is built twice from the same source: -O2 -mpopcnt, and -O3 -mavx2 (the reference's own flags,
        # ref CMakeLists.txt:112-148); both are timed on the final sample and the faster one is the one reported
        builds = {"": "-O2 -mpopcnt", "avx2": "-O3 -mavx2 -mbmi2 -mpopcnt"}
        timed = {"": dt}
        try:
            O.set_variant("avx2")
            run_sample(min(m, 2_000_000))  # load + warm
            timed["avx2"], tl = run_sample(m)
        except OSError:
            pass
        variant = min(timed, key=timed.get)
        O.set_variant(variant)
        dt = timed[variant]
        # SURVEY.md 8d also asks for the 1-thread and 8-thread figures of the same port
        fixed = {}
        for t, mt in ((1, 2_000_000), (8, 8_000_000)):
            mt = min(sample_cap, mt)
            dtt, _ = run_sample(mt, t)
            fixed[f"threads_{t}"] = {"value": round(mt / dtt / 1e6, 3), "sample": mt}
        oversubscribed = None
        if over != cores:
            dto, _ = run_sample(m, over)
            oversubscribed = {"threads": over, "value": round(m / dto / 1e6, 3), "sample": m,
                              "what": f"the same sample on {over} threads over the {granted} granted CPUs (the path is DRAM-latency bound)"}
        cpu = {"value": round(m / dt / 1e6, 3), "unit": "Mkmers/s", "cores": cores, "threads": cores, "cpus_granted": granted,
               "cpus_of_the_box": os.cpu_count(), "oversubscribed": oversubscribed, "kind": "port", **fixed,
               "threads_1_value": fixed["threads_1"]["value"], "threads_8_value": fixed["threads_8"]["value"],  # flat copies (the 1- and 8-thread points)
               "build": builds[variant],
               "builds_timed_Mkmers_per_s": {builds[v]: round(m / t / 1e6, 3) for v, t in timed.items()},
               "sample": f"first {m} of the {Q} {args.workload} {kdesc} of rank 0, {args.mode}, same index, "
                         f"{dt:.1f} s wall on {cores} threads ({granted} CPUs granted of {os.cpu_count()}), "
                         f"results equal to the GPU's",
               "per_query": {"steps": round(tl["steps"] / m, 4), "distinct_blocks": round(tl["blocks"] / m, 4)}}

    # ---- host-inclusive rates (never `value`): the same batch from host memory through the flat packed pipeline
    # (awfmGpuStreamPacked: chunked upload / kernels / download) and a prefix of it through the drop-in
    # AoS entry point (awFmParallelSearchCount/Locate, ref src/AwFmParallelSearch.c:95-220), wall clock, rank 0 ----
    e2e = None
    if not args.no_e2e and world == 1 and d_offsets is None and K <= (12 if amino else 32):
        e2e = end_to_end(args, L, api, g, ix, d_chars, d_counts, d_hit_off, state, Q, K, amino, dev, d_planted, first, n)

    def time_piece(p, steps, force=None):
        """probe + one warm-up + `steps` timed steps (wall clock between two device synchronisations) of a piece; ms per step"""
        # (without the library's per-kernel events, which only the main timed loop needs: four events per search are ~20 us)
        timing = os.environ.pop("AWFM_GPU_TIME_ORDERED", None)
        try:
            probe(p, force)
            run_steps(p, len(lanes))
            # (the smaller of two timed loops: a loop of a few 0.4-ms steps is stalled now and then by something else on the box --
            # one of eight shards at 0.58 instead of 0.39 ms in a run of round 6 -- and the proxy takes the MAX over the shards)
            dt = None
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run_steps(p, steps)
                torch.cuda.synchronize()
                once = (time.perf_counter() - t0) / steps
                dt = once if dt is None else min(dt, once)
            check_against_probe(p, steps)
        finally:
            if timing is not None:
                os.environ["AWFM_GPU_TIME_ORDERED"] = timing
        return dt * 1e3

    # ---- the DENSE form of the results, timed beside whichever form the steps used (3 steps): range / count under every
    # k-mer number, hit offsets over the whole batch, positions in k-mer order -- the reference's result, flattened ----
    dense_form = None
    if args.dense_form and locate and ordered and whole.form != "dense" and not whole.windowed:
        p = Piece(0, Q)
        ms = time_piece(p, 3, force="dense")
        dense_form = {"ms_per_step": round(ms, 3), "value": round(Q / ms / 1e3, 1), "steps": 3, "hits": p.hits,
                      "timed_form": whole.form, "timed_form_ms_per_step": round(ms_per_step, 3)}
        assert p.hits == whole.hits, "the dense form finds another number of hits"

    # ---- secondary: the same index, 10^8 k-mers drawn from the text (BASELINE config 3b), same step, 3 timed steps ----
    secondary = None
    if d_planted is not None:
        from avxwindowfmindex_amd import synth

        planted = Piece(0, Q, chars=d_planted)
        dt = time_piece(planted, 3) * 1e-3
        hits = planted.hits
        planted_pos = to_dense(planted)
        # every planted k-mer must come back at its planting offset (checked on the first 10^6: a k-mer with one hit has
        # exactly that position, one with several has it among them)
        m = min(Q, 1_000_000)
        planted_at = synth.planted_offsets(103, m, K, n, first=first)
        ho = d_hit_off[: m + 1].cpu().numpy().view(np.uint64)
        pos = planted_pos[: int(ho[m])].cpu().numpy().view(np.uint64)
        cnt = np.diff(ho)
        assert cnt.min() >= 1, "a planted k-mer was not found"
        one = cnt == 1
        assert np.array_equal(pos[ho[:-1][one]], planted_at[one]), "a planted k-mer was located somewhere else"
        for i in np.flatnonzero(~one)[:1000]:
            assert planted_at[i] in pos[ho[i]:ho[i + 1]], "a planted k-mer's own offset is missing from its hit list"
        pkey = digest.key(args.alphabet, "planted", "locate", n, str(K), args.seed_k, args.sa_ratio, first, Q)
        pdig = {"counts": f"{digest.counts_digest(first, d_hit_off[1:] - d_hit_off[:-1]):016x}",
                "positions": f"{digest.positions_digest(first, d_hit_off, planted_pos[:hits]):016x}"}
        committed = digest.load_golden().get(pkey)
        assert committed is None or committed == pdig, f"planted digests {pdig} differ from the committed {committed}"
        del planted_pos, pos
        # the dense form of the same batch (see dense_form above)
        pd = Piece(0, Q, chars=d_planted)
        planted_dense_ms = time_piece(pd, 3, force="dense")
        assert pd.hits == hits
        # the same steps the OTHER way of locating: through the device-only full suffix array (awfmGpuIndexSetDenseSa: 4 bytes
        # per BWT position of HBM, a locate is one gather) when the steps above walked, through the LF walk and the sampled
        # array (the reference's backtrace) when the image carries the full array: identical positions either way
        other = None
        try:
            t1 = time.perf_counter()
            g.set_dense_sa(not dense_sa_default)
            torch.cuda.synchronize()
            other_build = time.perf_counter() - t1
            other_dt = time_piece(planted, 3) * 1e-3
            other_positions = to_dense(planted)
            other_pos = f"{digest.positions_digest(first, d_hit_off, other_positions[:hits]):016x}"
            del other_positions
            assert other_pos == pdig["positions"], "positions through the full suffix array differ from the walk's"
            other = {"value": round(Q / other_dt / 1e6, 2), "ms_per_step": round(other_dt * 1e3, 3),
                     "checked": "positions digest equals the other way's"}
            if not dense_sa_default:
                other.update(build_s=round(other_build, 2), extra_device_bytes=4 * ix.bwt_length)
        except api.AwFmError as e:  # not enough device memory: the line simply has no such entry
            other = {"skipped": str(e)}
        g.set_dense_sa(dense_sa_default)
        torch.cuda.synchronize()
        secondary = {"workload": f"{Q / 1e6:g} M planted {K}-mers (every k-mer has >= 1 hit), locate, same index",
                     "device_dense_sa": dense_sa_default,
                     ("with_lf_walk" if dense_sa_default else "with_device_dense_sa"): other,
                     "value": round(Q / dt / 1e6, 2), "unit": "Mkmers/s", "ms_per_step": round(dt * 1e3, 3), "steps": 3,
                     "result_form": planted.form,
                     "dense_form": {"ms_per_step": round(planted_dense_ms, 3), "value": round(Q / planted_dense_ms / 1e3, 1), "steps": 3},
                     "hits_per_step": int(hits), "checked": f"first {m} k-mers located at their planting offsets",
                     "digests": dict(pdig, status="match" if committed else "unknown")}
        if args.record_digests:
            known = json.load(open(args.record_digests)) if os.path.exists(args.record_digests) else {}
            known[pkey] = pdig
            json.dump(known, open(args.record_digests, "w"), indent=1, sort_keys=True)

    # ---- and cfg 5's mixed-length batch on the same index, counted (awfmGpuSearchHits with CSR offsets): 3 timed steps,
    # the counts' digest against the committed one of `--workload mixed` ----
    if d_mixed is not None and secondary is not None:
        m_chars, m_off = d_mixed
        m_counts = torch.empty(Q, dtype=torch.int32, device=dev)
        lane0 = lanes[0]

        def mixed_step():
            g.search_hits(m_chars.data_ptr(), m_off.data_ptr(), 0, Q, 0, m_counts.data_ptr(), lane0.stream)
        torch.cuda.synchronize()
        mixed_step()  # (the first mixed-length batch builds the tables per k-mer length)
        mixed_step()
        torch.cuda.synchronize()
        g.ordered_kernel_log()  # (drop the log entries of everything before)
        t0 = time.perf_counter()
        for _ in range(3):
            mixed_step()
        torch.cuda.synchronize()
        mixed_dt = (time.perf_counter() - t0) / 3
        mixed_lookup_chosen = bool(g.last_ordered_kernel_is_lookup())
        mlog = g.ordered_kernel_log()
        mkey = digest.key(args.alphabet, "mixed", "count", n, "8-30", args.seed_k, args.sa_ratio, first, Q)
        mdig = f"{digest.counts_digest(first, m_counts):016x}"
        committed = digest.load_golden().get(mkey)
        assert committed is None or committed["counts"] == mdig, f"mixed-length counts digest {mdig} differs from the committed {committed}"
        lt_bytes, lt_s = g.length_tables
        secondary["mixed_lengths"] = {
            "workload": f"{Q / 1e6:g} M k-mers of 8..30 characters (every second one drawn from the text), counted, same index",
            "value": round(Q / mixed_dt / 1e6, 2), "unit": "Mkmers/s", "ms_per_step": round(mixed_dt * 1e3, 3), "steps": 3,
            "kernel": "mixedLookupSearchKernel (one table entry per k-mer: the table of its own length, or the deeper table)"
                      if mixed_lookup_chosen else "orderedSearchKernel (16-byte records)",
            "kernel_ms": round(float(np.mean([(f if mixed_lookup_chosen else k) for f, k in mlog])), 3) if mlog else None,
            "length_tables_bytes": lt_bytes, "length_tables_build_s": round(lt_s, 3),
            "kmers_with_hits": int((m_counts != 0).sum().item()),
            "digests": {"counts": mdig, "status": "match" if committed else "unknown"}}
        del m_counts, m_chars, m_off, d_mixed

    # ---- and BASELINE configs[3] -- 5 * 10^7 random amino 10-mers against a Swiss-Prot-sized index (2 * 10^8 residues), located --
    # with its own index, in this same run; the HBM-bound variant (2 * 10^9 residues: the image is ten times the Infinity Cache)
    # as well while the run is young enough ----
    if secondary is not None and not args.no_amino:
        secondary["amino"] = amino_leg(L, api, digest, torch, np, dev, 200_000_000, record_digests=args.record_digests)
        if time.time() - T_START < 150:
            secondary["amino_2e9"] = amino_leg(L, api, digest, torch, np, dev, 2_000_000_000, steps=3, record_digests=args.record_digests)

    # ---- strong-scaling proxy on ONE GPU: the contiguous shards N ranks would hold of this batch (configs[2]: "query
    # batch sharded 1 -> 8"), each timed by itself with the same step; a rank of an N-GPU run does exactly this work on
    # its own replica, with nothing exchanged (ref src/AwFmParallelSearch.c:103-129: 8-query blocks are independent), so
    # N x the slowest shard's time is what N GPUs would take.  The shards' digests must add up to the batch's. ----
    proxy = None
    if args.shard_proxy and world == 1 and not whole.windowed:
        whole_again_ms = time_piece(Piece(0, Q), args.proxy_steps)  # the whole batch, timed the way the shards are
        proxy = {"what": "the N contiguous shards of this batch, each timed alone on this GPU (probe + warm-up + steps, wall "
                         "clock between device synchronisations, no per-kernel events); efficiency = whole-batch ms timed the "
                         "same way / (N x slowest shard's ms).  A step's cost is a + b x k-mers: b is the table gather "
                         "(encodeLookupKernel, 26 us per 10^6 k-mers), a the chains that do not shrink with the batch -- the "
                         "search of the k-mers kept (3 dependent block reads x 5 rounds of what fits the chip), the longest LF "
                         "walk of the batch (~60 steps), and ~20 launches",
                 "whole_batch_ms": round(whole_again_ms, 4), "whole_batch_ms_timed_loop": round(ms_per_step, 4),
                 "steps": args.proxy_steps, "shards": {}}
        shard_digests = {}
        sources = [("batch", d_chars, mine)]
        if d_planted is not None:
            sources.append(("planted", d_planted, (first, Q, int(pdig["counts"], 16), int(pdig["positions"], 16))))
            planted_whole_ms = time_piece(Piece(0, Q, chars=d_planted), args.proxy_steps)
            proxy["planted_whole_batch_ms"] = round(planted_whole_ms, 4)
        for name, chars, full in sources:
            per_n = {}
            for parts in (2, 4, 8):
                times, sum_c, sum_p = [], 0, 0
                for r in range(parts):
                    if d_offsets is not None:
                        lo, hi = shard.balanced_bounds(d_offsets, parts, r)
                    else:
                        lo, hi = shard.shard_bounds(Q, parts, r)
                    p = Piece(lo, hi - lo, chars=chars)
                    times.append(time_piece(p, args.proxy_steps))
                    if locate:
                        ppos = to_dense(p)
                        dc = digest.counts_digest(first + lo, d_hit_off[1:p.q + 1] - d_hit_off[:p.q])
                        dp = digest.positions_digest(first + lo, d_hit_off[: p.q + 1], ppos[: max(p.hits, 1)])
                        del ppos
                    else:
                        dc, dp = digest.counts_digest(first + lo, d_counts[: p.q]), None
                    sum_c += dc
                    sum_p += dp or 0
                    # the shard's own committed digest (what rank r of an N-rank strong run must produce), when there is one
                    skey = digest.key(args.alphabet + ("" if args.text == "uniform" else "-" + args.text), "planted" if name == "planted" else args.workload,
                                      args.mode, n, kmer_name, args.seed_k, args.sa_ratio, first + lo, hi - lo)
                    sdig = {"counts": f"{dc:016x}", "positions": f"{dp:016x}" if dp is not None else None}
                    committed_shard = digest.load_golden().get(skey)
                    assert committed_shard is None or committed_shard == sdig, f"shard digest {sdig} differs from the committed {committed_shard} ({skey})"
                    shard_digests[skey] = sdig
                assert (sum_c & digest.MASK) == full[2], f"{name}: the counts digests of {parts} shards do not add up to the batch's"
                assert full[3] is None or (sum_p & digest.MASK) == full[3], f"{name}: the positions digests of {parts} shards do not add up"
                base_ms = whole_again_ms if name == "batch" else planted_whole_ms
                per_n[str(parts)] = {"kmers_per_shard": Q // parts, "ms_max": round(max(times), 4), "ms_mean": round(float(np.mean(times)), 4),
                                     "Mkmers_per_s_at_N_gpus": round(Q / max(times) / 1e3, 1),
                                     "efficiency": round(base_ms / (parts * max(times)), 4)}
            proxy["shards"][name] = per_n
        # ---- round 6: the dense-hit batch sharded by SEED BUCKET instead of by batch position (include/awfm_gpu.h:
        # awfmGpuOrderKmers / awfmGpuSearchOrderedRecords; dist.bucket_exchange is the exchange N real ranks run).  A rank's
        # step: order its own contiguous N-th of the batch (timed), send every other rank the records of that rank's buckets
        # and receive its own (PRICED: one GPU cannot time an exchange over xGMI), put the N slices it received in bucket order
        # (awfmGpuMergeBucketRuns, timed, on the slices the N shards' own orders give), search the dense N-th of the ORDER it
        # then holds and locate its hits (timed, on those merged records).  Results carry the k-mers' numbers in the whole batch: the ranks' digests add up to the batch's. ----
        if d_planted is not None and locate and d_offsets is None and g.order_buckets(K, Q):
            buckets = g.order_buckets(K, Q)
            d_all_recs = torch.empty(Q, dtype=torch.int64, device=dev)
            d_all_bs = torch.empty(buckets + 3, dtype=torch.int32, device=dev)
            st = lanes[0].stream
            g.order_kmers(d_planted.data_ptr(), K, Q, first, first + Q, d_all_recs.data_ptr(), d_all_bs.data_ptr(), st)
            torch.cuda.synchronize()
            all_bs = d_all_bs.cpu().to(torch.int64)
            assert int(all_bs[buckets + 2]) == 0, "planted k-mers have no ambiguity characters"
            XGMI_LINK_GBS, XGMI_EFFICIENCY = 153.0, 0.7  # per link and direction (MI355X: 7 links per GPU); what a large all-to-all sustains: assumed
            per_n = {}
            for parts in (2, 4, 8):
                cuts = shard.bucket_cuts(buckets, parts)
                # the orders of all N contiguous shards (what the N ranks would hold before the exchange), made once
                shard_recs, shard_bs = [], []
                for j in range(parts):
                    lo, hi = shard.shard_bounds(Q, parts, j)
                    recs_j = torch.empty(hi - lo, dtype=torch.int64, device=dev)
                    bs_j = torch.empty(buckets + 3, dtype=torch.int32, device=dev)
                    g.order_kmers(d_planted.data_ptr() + lo * K, K, hi - lo, first + lo, first + Q, recs_j.data_ptr(), bs_j.data_ptr(), st)
                    shard_recs.append(recs_j)
                    shard_bs.append(bs_j)
                torch.cuda.synchronize()
                bs_host = [b.cpu().to(torch.int64) for b in shard_bs]
                rows, sum_c, sum_p = [], 0, 0
                for r in range(parts):
                    lo, hi = shard.shard_bounds(Q, parts, r)
                    d_recs = torch.empty(hi - lo, dtype=torch.int64, device=dev)
                    d_bs = torch.empty(buckets + 3, dtype=torch.int32, device=dev)

                    def order_own():
                        g.order_kmers(d_planted.data_ptr() + lo * K, K, hi - lo, first + lo, first + Q, d_recs.data_ptr(), d_bs.data_ptr(), st)

                    # what rank r receives: slice j = the records of r's buckets in shard j's order, with the starts of its buckets
                    c0, c1 = cuts[r], cuts[r + 1]
                    sizes = [int(bs_host[j][c1] - bs_host[j][c0]) for j in range(parts)]
                    d_received = torch.cat([shard_recs[j][int(bs_host[j][c0]): int(bs_host[j][c1])] for j in range(parts)])
                    d_slice_at = torch.tensor([sum(sizes[:j]) for j in range(parts)], dtype=torch.int64).to(dev)
                    d_starts = torch.stack([(bs_host[j][c0: c1 + 1] - bs_host[j][c0]).to(torch.int32) for j in range(parts)]).to(dev)
                    m = int(sum(sizes))
                    assert m == int(all_bs[c1]) - int(all_bs[c0]), "the slices of a rank's buckets do not add up to those buckets of the whole order"
                    d_mine = torch.empty(m, dtype=torch.int64, device=dev)
                    d_full = torch.empty(buckets + 3, dtype=torch.int32, device=dev)

                    def merge_own():
                        g.merge_bucket_runs(d_received.data_ptr(), d_slice_at.data_ptr(), d_starts.data_ptr(), parts, c0, c1, buckets,
                                            d_mine.data_ptr(), d_full.data_ptr(), st)

                    merge_own()
                    d_k = torch.empty(m, dtype=torch.int32, device=dev)
                    d_r = torch.empty(m * 2, dtype=torch.int64, device=dev)
                    d_o = torch.empty(m + 1, dtype=torch.int64, device=dev)
                    d_sc = torch.empty(api.GpuIndex.scan_scratch_bytes(m), dtype=torch.uint8, device=dev)
                    g.search_ordered_records(d_mine.data_ptr(), d_full.data_ptr(), c0, c1, K, first + Q, d_k.data_ptr(), d_r.data_ptr(), st)
                    g.hit_offsets_on_device(0, d_r.data_ptr(), m, d_o.data_ptr(), d_sc.data_ptr(), st)
                    torch.cuda.synchronize()
                    hits_r = int(d_o[m].item())
                    d_p = torch.empty(hits_r + hits_r // 8 + 64, dtype=torch.int64, device=dev)

                    d_c = torch.empty(m, dtype=torch.int32, device=dev) if narrow_counts else None  # (counts in search order: what the scan reads)

                    def search_own():
                        g.search_ordered_records(d_mine.data_ptr(), d_full.data_ptr(), c0, c1, K, first + Q, d_k.data_ptr(), d_r.data_ptr(), st,
                                                 d_order_counts=d_c.data_ptr() if narrow_counts else 0)
                        if narrow_counts:
                            g.hit_offsets_on_device(d_c.data_ptr(), 0, m, d_o.data_ptr(), d_sc.data_ptr(), st)
                        else:
                            g.hit_offsets_on_device(0, d_r.data_ptr(), m, d_o.data_ptr(), d_sc.data_ptr(), st)
                        g.locate_on_device(d_r.data_ptr(), d_o.data_ptr(), m, d_p.numel(), d_p.data_ptr(), st)

                    def timed(fn):  # (the smaller of two timed loops, as time_piece)
                        fn()
                        best = None
                        for _ in range(2):
                            torch.cuda.synchronize()
                            t1 = time.perf_counter()
                            for _ in range(args.proxy_steps):
                                fn()
                            torch.cuda.synchronize()
                            once = (time.perf_counter() - t1) * 1e3 / args.proxy_steps
                            best = once if best is None else min(best, once)
                        return best

                    order_ms = timed(order_own)
                    merge_ms = timed(merge_own)
                    search_ms_r = timed(search_own)
                    if os.environ.get("AWFM_BENCH_SEED_BUCKET_PROBE") and r == 0:  # where a rank's step goes, and the same on the whole batch's order
                        probe = {"search": timed(lambda: g.search_ordered_records(d_mine.data_ptr(), d_full.data_ptr(), c0, c1, K, first + Q, d_k.data_ptr(), d_r.data_ptr(), st)),
                                 "offsets": timed(lambda: g.hit_offsets_on_device(0, d_r.data_ptr(), m, d_o.data_ptr(), d_sc.data_ptr(), st)),
                                 "locate": timed(lambda: g.locate_on_device(d_r.data_ptr(), d_o.data_ptr(), m, d_p.numel(), d_p.data_ptr(), st)),
                                 "search_whole_order": timed(lambda: g.search_ordered_records(d_all_recs.data_ptr(), d_all_bs.data_ptr(), c0, c1, K, first + Q, d_k.data_ptr(), d_r.data_ptr(), st))}
                        g.search_ordered_records(d_all_recs.data_ptr(), d_all_bs.data_ptr(), c0, c1, K, first + Q, d_k.data_ptr(), d_r.data_ptr(), st)
                        g.hit_offsets_on_device(0, d_r.data_ptr(), m, d_o.data_ptr(), d_sc.data_ptr(), st)
                        probe["locate_whole_order"] = timed(lambda: g.locate_on_device(d_r.data_ptr(), d_o.data_ptr(), m, d_p.numel(), d_p.data_ptr(), st))
                        sys.stderr.write(f"[seed-bucket probe] N={parts}: {probe}\n")
                        search_own()
                        torch.cuda.synchronize()
                    # every rank sends (and receives) N - 1 slices of about m / N records at once, each over a link of its own
                    exchange_ms = (m / parts * 8) / (XGMI_LINK_GBS * XGMI_EFFICIENCY * 1e9) * 1e3 if parts > 1 else 0.0
                    rows.append({"order_own_shard_ms": round(order_ms, 4), "exchange_ms_priced": round(exchange_ms, 4), "merge_ms": round(merge_ms, 4),
                                 "search_and_locate_ms": round(search_ms_r, 4), "kmers": m, "hits": hits_r,
                                 "total_ms": round(order_ms + exchange_ms + merge_ms + search_ms_r, 4),
                                 # two batches in flight (bench.py --sharding seed_bucket runs that way): the exchange of batch i + 1 is
                                 # in the links while batch i is searched, so a step costs what the GPU does, or the exchange if longer
                                 "pipelined_ms": round(max(order_ms + merge_ms + search_ms_r, exchange_ms), 4)})
                    ids = d_k.to(torch.int64)
                    assert int(ids.min().item()) >= first and int(ids.max().item()) < first + Q
                    dc_r = digest.counts_digest_keyed(ids, d_o[1:] - d_o[:-1]) & digest.MASK
                    dp_r = digest.positions_digest_keyed(ids, d_o, d_p[: max(hits_r, 1)]) & digest.MASK
                    sum_c += dc_r
                    sum_p += dp_r
                    # the rank's own committed digest (what rank r of N must produce when the batch is cut by seed bucket)
                    rkey = digest.key(args.alphabet, f"planted_seed_bucket_rank{r}_of_{parts}", args.mode, n, kmer_name, args.seed_k, args.sa_ratio, first, Q)
                    rdig = {"counts": f"{dc_r:016x}", "positions": f"{dp_r:016x}", "kmers": m}
                    committed_rank = digest.load_golden().get(rkey)
                    assert committed_rank is None or committed_rank == rdig, f"seed-bucket rank digest {rdig} differs from the committed {committed_rank} ({rkey})"
                    shard_digests[rkey] = rdig
                    del d_recs, d_bs, d_k, d_r, d_o, d_sc, d_p, d_mine, d_full, d_received, ids
                del shard_recs, shard_bs
                assert (sum_c & digest.MASK) == int(pdig["counts"], 16), f"seed-bucket sharding: the counts digests of {parts} ranks do not add up to the batch's"
                assert (sum_p & digest.MASK) == int(pdig["positions"], 16), f"seed-bucket sharding: the positions digests of {parts} ranks do not add up"
                slowest = max(rw["total_ms"] for rw in rows)
                slowest_pipelined = max(rw["pipelined_ms"] for rw in rows)
                per_n[str(parts)] = {"sharding": "seed_bucket", "ms_max": round(slowest, 4), "ranks": rows,
                                     "Mkmers_per_s_at_N_gpus": round(Q / slowest / 1e3, 1), "efficiency": round(planted_whole_ms / (parts * slowest), 4),
                                     "ms_max_two_batches_in_flight": round(slowest_pipelined, 4),
                                     "efficiency_two_batches_in_flight": round(planted_whole_ms / (parts * slowest_pipelined), 4)}
            proxy["shards"]["planted_seed_bucket"] = per_n
            proxy["seed_bucket_note"] = ("planted_seed_bucket: every rank orders its contiguous shard (awfmGpuOrderKmers), exchanges records by bucket range "
                                         f"(priced at {XGMI_LINK_GBS:g} GB/s per xGMI link x {XGMI_EFFICIENCY:g}: one GPU cannot time it), and searches the dense N-th of the "
                                         "ORDER (awfmGpuSearchOrderedRecords); digests keyed by the k-mers' numbers in the whole batch add up to the batch's.  "
                                         "`efficiency`: one batch at a time, the exchange exposed; `efficiency_two_batches_in_flight`: the throughput of a stream of "
                                         "batches the way `--sharding seed_bucket` runs them (the exchange of batch i + 1 behind the search of batch i, on its own "
                                         "stream) -- a MODEL on one GPU: it assumes the all-to-all costs the kernels nothing")
            del d_all_recs, d_all_bs
        proxy["digests"] = ("the shards' counts and positions digests add up to the whole batch's for every N; every shard that has "
                            "a committed digest of its own (tests/golden/bench_digests.json) equals it")
        if args.record_digests:
            known = json.load(open(args.record_digests)) if os.path.exists(args.record_digests) else {}
            known.update(shard_digests)
            json.dump(known, open(args.record_digests, "w"), indent=1, sort_keys=True)
    del d_planted
    # ---- the same step on the text shape real users have: a genome-shaped 3.1 Gbp text with an index of its own, 10^8 21-mers
    # drawn from its unique sequence, located (repetitive_leg); while the run is young enough ----
    if secondary is not None and not args.no_repetitive and time.time() - T_START < 240:
        torch.cuda.empty_cache()
        secondary["repetitive"] = repetitive_leg(L, api, digest, torch, np, dev, record_digests=args.record_digests)

    deep_last_build = g.deep_seed_build
    # ---- the drop-in user's FIRST call: the index in host memory (as awFmReadIndexFromFile leaves it), no device image yet.
    # awFmParallelSearchLocate then uploads the image (3.8 GB), builds the pair image and the deeper table, and searches.
    # The image of this run is dropped for it, so this is the last thing the run does with the GPU. ----
    image_bytes, image_deep_k = g.device_bytes, g.deep_seed_k or args.seed_k
    image_described = g.describe()  # which accelerators the image holds, which ones it did not get and why
    length_tables_built = g.length_tables  # (bytes, seconds): before the image is released for the first-call leg
    first_call = None
    if e2e is not None and locate and args.e2e_aos_queries:
        import ctypes as C
        m = min(Q, 1_000_000)
        chars = np.ascontiguousarray(d_chars[: m * K].cpu().numpy())
        lst = api.KmerSearchList(m)
        arr = np.ctypeslib.as_array(C.cast(lst.ptr.contents.kmerSearchData, C.POINTER(C.c_uint64)), shape=(m, 4))
        arr[:, 0] = chars.ctypes.data + np.arange(m, dtype=np.uint64) * np.uint64(K)
        arr[:, 1] = K
        lst.ptr.contents.count = m
        threads = min(32, 2 * (os.cpu_count() or 1))
        torch.cuda.synchronize()
        g.handle = None  # the handle belongs to the index's registry entry, which goes now
        L.awfmGpuIndexRelease(ix.ptr)
        t0 = time.perf_counter()
        rc = api.parallel_search_locate(ix, lst, threads)
        t1 = time.perf_counter()
        api.parallel_search_locate(ix, lst, threads)
        t2 = time.perf_counter()
        assert rc == api.AwFmSuccess
        got = np.ctypeslib.as_array(C.cast(lst.ptr.contents.kmerSearchData, C.POINTER(C.c_uint32)), shape=(m, 8))[:, 6]
        assert int(got.sum()) > 0
        again = api.GpuIndex(ix, acquire=True)  # the image that call made, complete (awfmGpuIndexAcquire waits for what is built behind the searches)
        t3 = time.perf_counter()
        api.parallel_search_locate(ix, lst, threads)
        t4 = time.perf_counter()
        rebuilt_deep_s = again.deep_seed_build[0]
        first_call = {"first_call_s": round(t1 - t0, 3), "second_call_s": round(t2 - t1, 4), "kmers": m,
                      "accelerators_installed_after_s": round(t3 - t0, 3), "call_through_the_complete_image_s": round(t4 - t3, 4),
                      "image_bytes": again.device_bytes, "image_deep_seed_k": again.deep_seed_k, "image_dense_sa": again.has_dense_sa,
                      "deep_table_s_in_that_call": round(rebuilt_deep_s, 3),
                      # (a fresh process that reads the index from its file and makes this call: scripts/first_call_probe.py)
                      "what": "awFmParallelSearchLocate on an index that has no device image yet: image upload + pair image + the search "
                              "(round 6: the deeper table and the full suffix array are built by a thread of their own behind the first calls and "
                              "installed between two calls; accelerators_installed_after_s: when awfmGpuIndexAcquire hands the complete image "
                              "over); the second call is the same list again, while they are being built"}
        again.handle = None
        if e2e is not None:
            e2e["first_call"] = first_call
        lst.dealloc()
        g = None

    # ---- round 6: the same steps on an index BEYOND 2^32 positions (6.2 Gbp: a two-strand human genome's size), built here once
    # this run's own image is gone -- the 64-bit suffix sort takes 33 bytes of HBM per text position at its peak (wide_leg) ----
    if secondary is not None and not args.no_wide and not amino and world == 1 and n < (1 << 32) and time.time() - T_START < 400:
        if g is not None:
            g.handle = None
            g = None
        L.awfmGpuIndexRelease(ix.ptr)
        pos_buf["t"] = None
        for ln in lanes:
            ln.pos = None
        torch.cuda.empty_cache()
        from avxwindowfmindex_amd import synth as _synth
        secondary["wide"] = wide_leg(L, api, digest, _synth, torch, np, dev, record_digests=args.record_digests)

    # the clocks this box runs at (boxes of the pool differ by several per cent on the same code: the line says which one it
    # was).  Read from sysfs -- no child process: a process that has initialised the GPU must not start one.
    clocks = None
    try:
        import glob
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        if cards:
            base = os.path.dirname(cards[min(max(args.force_device, 0) if world == 1 else local_rank, len(cards) - 1)])
            clocks = {}
            for name in ("sclk", "mclk", "fclk"):
                path = os.path.join(base, f"pp_dpm_{name}")
                if os.path.exists(path):
                    levels = open(path).read().split("\n")
                    current = [ln for ln in levels if ln.strip().endswith("*")]
                    clocks[name] = (current[0].split(":")[1].replace("*", "").strip() if current else None)
                    clocks[name + "_max"] = levels[-2].split(":")[1].replace("*", "").strip() if len(levels) > 1 and ":" in levels[-2] else None
    except Exception:  # noqa: BLE001  (reporting only)
        clocks = None
    per = "per GPU" if args.scaling == "weak" else f"in total, sharded over {world} rank(s)"
    deep_build_s, deep_transient = deep_first_build if args.device_seed_k < 0 else deep_last_build
    deep_rebuild_s = deep_last_build[0]
    form_names = {"order": "every k-mer {k-mer number, range} in search order + hit offsets and positions in that order",
                  "list": "list of the k-mers with hits {k-mer number, range} in k-mer order + hit offsets over the list + positions",
                  "dense": "range / count under every k-mer number + hit offsets over the batch + positions"}
    config = {"workload": f"{args.queries / 1e6:g} M {args.workload} {kdesc} {per}, {args.mode}, "
                          f"{n / 1e9:g} G{'res' if amino else 'bp'} {args.text} synthetic {args.alphabet} text"
                          f"{' (GRCh38-sized)' if n >= 3_000_000_000 and not amino else ''}, "
                          f"SA ratio {args.sa_ratio}, seed table k={args.seed_k}",
              "parallelism": f"index replica per GPU, query batch sharded over {world} rank(s), no collective",
              "batch_kmers": batch_total, "rank0_kmers": Q,
              "timing_collective": shard.timing_backend() or "none (one rank)",
              "hits_per_step_rank0": int(state["hits"]), "locate_kernels_ms": round(locate_ms, 3),
              "search_call_ms": round(search_ms, 3), "host_waits_per_step": 0 if not whole.windowed else "one per window",
              # consecutive steps alternate between this many streams, each with its own result buffers and scratch slot
              "streams": 1 if whole.windowed else len(lanes),
              "index_build_s": round(build_s, 2),
              "device_image_bytes": image_bytes, "device_seed_k": image_deep_k, "device_image": image_described,
              # the deeper table's construction, whoever started it (the library by itself at awfmGpuIndexAcquire, inside
              # index_build_s; or --device-seed-k): wall seconds and the device memory held beyond the table at the peak.  In
              # this process it follows the GPU index builder; 0.55 s in most runs, 4-6 s in some -- one hipMalloc, the first the
              # runtime cannot serve from blocks it holds (memory to hand back or to scrub after the process before); in a process
              # that reads its index from a file the construction takes 0.6 s (scripts/first_call_probe.py; device_seed_rebuild_s here)
              "device_seed_build_s": round(deep_build_s, 2), "device_seed_transient_bytes": int(deep_transient),
              # of which inside the hipMalloc calls (the table's 34 GB): the seconds some runs show are the allocation, not the kernels
              "device_seed_alloc_s": round(deep_first_alloc_s, 2),
              "device_seed_rebuild_s": round(deep_rebuild_s, 2),  # the same construction once more (after roofline_general dropped the table): the allocator has the memory at hand
              "device_dense_sa": dense_sa_default, "device_dense_sa_build_s": round(dense_s, 2),
              "search_path": ({"order": "awfmGpuSearchHitsInOrder", "list": "awfmGpuSearchHitsCompact",
                               "dense": "awfmGpuSearchHitsSparse" if narrow_counts else "awfmGpuSearchHits"}[whole.form] if locate
                              else "awfmGpuSearchHits") + (", seed order" if ordered else ", mixedLookupSearchKernel" if small_mixed_lookup else ", general kernel"),
              "result_format": form_names[whole.form] if locate else "count under every k-mer number",
              "result_format_probe_ms": getattr(whole, "form_probe_ms", None),  # mixed-length dense-hit batches: one search call of each form, timed by the probe
              "lookup_front": {0: "both front ends launched, the sample decides on the device", 1: "lookup kernel only (predicted from an earlier step's sample)",
                               2: "ordered kernels only (predicted from an earlier step's sample)"}.get(lookup_front, "no sample"),
              "list_tail": "awfmGpuListLocateOnDevice (one launch)" if locate and whole.form == "list" and list_tail else None}
    lt_bytes, lt_s = length_tables_built
    if lt_bytes:  # built by the probe step of a mixed-length batch (awfm_mixed_lookup_kernel.h): device-only, kept with the image
        config["length_tables_bytes"] = lt_bytes
        config["length_tables_build_s"] = round(lt_s, 3)
    if dense_form:  # flat copies for readers that keep scalars only
        config["dense_form"] = dense_form
        config["dense_form_ms_per_step"] = dense_form["ms_per_step"]
        config["dense_form_value"] = dense_form["value"]
    if secondary:
        config["planted_ms_per_step"] = secondary["ms_per_step"]
        if secondary.get("with_lf_walk") and "ms_per_step" in secondary["with_lf_walk"]:
            config["planted_lf_walk_ms_per_step"] = secondary["with_lf_walk"]["ms_per_step"]
        config["planted_dense_form_ms_per_step"] = secondary["dense_form"]["ms_per_step"]
        if "mixed_lengths" in secondary:
            config["mixed_lengths_ms_per_step"] = secondary["mixed_lengths"]["ms_per_step"]
            config["mixed_lengths_value"] = secondary["mixed_lengths"]["value"]
        for name in ("amino", "amino_2e9", "repetitive"):
            if name in secondary:
                config[name + "_value"] = secondary[name]["value"]
                config[name + "_ms_per_step"] = secondary[name]["ms_per_step"]
                config[name + "_roofline_frac"] = secondary[name]["roofline"].get("frac")
        if "wide" in secondary:
            config["wide_value"] = secondary["wide"]["random"]["value"]
            config["wide_ms_per_step"] = secondary["wide"]["random"]["ms_per_step"]
            config["wide_roofline_frac"] = secondary["wide"]["random"]["roofline"]["frac"]
            config["wide_planted_value"] = secondary["wide"]["planted"]["value"]
            config["wide_planted_ms_per_step"] = secondary["wide"]["planted"]["ms_per_step"]
    if clocks:
        config["gpu_clocks"] = clocks
        for name in ("sclk", "mclk", "fclk"):
            if name in clocks:
                config[f"gpu_{name}"] = str(clocks[name])
    if kernel_log:
        doms = [(f if lookup_first else k) for f, k in kernel_log]
        config["dominant_kernel_ms_first_min_max"] = [round(doms[0], 3), round(min(doms), 3), round(max(doms), 3)]
    if first_call:
        config["aos_first_call_s"] = first_call["first_call_s"]
    if proxy:
        e8 = proxy["shards"]["batch"]["8"]
        config["scaling_proxy_8_efficiency"] = e8["efficiency"]
        if "planted" in proxy["shards"]:
            config["scaling_proxy_8_planted_efficiency"] = proxy["shards"]["planted"]["8"]["efficiency"]
        if "planted_seed_bucket" in proxy["shards"]:
            config["scaling_proxy_8_planted_seed_bucket_efficiency"] = proxy["shards"]["planted_seed_bucket"]["8"]["efficiency"]
            config["scaling_proxy_8_planted_seed_bucket_efficiency_two_batches_in_flight"] = proxy["shards"]["planted_seed_bucket"]["8"]["efficiency_two_batches_in_flight"]
        config["scaling_proxy_8_ms"] = e8["ms_max"]
    config["bench_wall_s"] = round(time.time() - T_START, 1)  # this process from its first line to its JSON line (imports, index, every leg)
    out = {
        "metric": "Mkmers/sec located, GRCh38 nucleotide index" if not amino else "Mkmers/sec located, amino index",
        "value": round(value, 2), "unit": "Mkmers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        # the same metric for k-mers that OCCUR (drawn from the text: every one located), which is what a seed-and-extend caller
        # sends -- the headline's random 21-mers almost never occur, and lookup first is at its best on them
        "value_present_kmers": secondary["value"] if secondary and "value" in secondary else None,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": config,
        "roofline": roofline,
        "roofline_general": roofline_general,
        "cpu_baseline": cpu,
        "digests": digest_check,
        "end_to_end": e2e,
        "secondary": secondary,
        "scaling_proxy": proxy,
    }
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
