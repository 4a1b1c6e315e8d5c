"""bench.py --gpus N starts N ranks by itself (no outer torchrun), every rank searches its own contiguous shard
with the product library, and the line it prints says n_gpus == N.

The GPU test runs two ranks on one GPU (gloo for the barrier / MAX reduction, --force-device 0) and checks
BOTH ranks' shard results -- ranges, hit offsets, positions in BWT order -- against the oracle on queries
regenerated on the host from the same seeds.  The shards are independent exactly like the reference's
8-query blocks (ref src/AwFmParallelSearch.c:103-129)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_launcher_reports_failure_of_its_ranks():
    """no GPU here: both ranks fail loudly, and so does the launcher (no silent single-rank run)"""
    from avxwindowfmindex_amd import _lib
    if _lib.lib().awfmGpuDeviceCount() > 0:
        pytest.skip("a GPU is visible: covered by the gpu test")
    r = _run_bench(["--gpus", "2", "--dist-backend", "gloo", "--force-device", "0", "--text-len", "3e5", "--queries",
                    "1e4", "--no-cpu"])
    assert r.returncode != 0
    assert "rank exit codes" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_must_match_gpus_flag():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE is 1" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("workload,mode", [("planted", "locate"), ("random", "count")])
def test_two_ranks_on_one_gpu_search_their_own_shards(oracle, awfm, require_gpu, tmp_path, workload, mode):
    from avxwindowfmindex_amd import synth
    n, Q, K, seed_k = 3_000_000, 1_000_000, 21 if workload == "planted" else 13, 8
    r = _run_bench(["--gpus", "2", "--force-device", "0", "--dist-backend", "gloo", "--text-len", "3e6", "--queries",
                    "1e6", "--kmer", str(K), "--seed-k", str(seed_k), "--workload", workload, "--mode", mode, "--no-cpu",
                    "--steps", "3", "--warmup", "1", "--scaling", "weak", "--streams", "2", "--dump-dir", str(tmp_path)])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    # ms_per_step is printed with three decimals: at a fraction of a millisecond per step that rounding alone is 0.2 %
    assert line["value"] == pytest.approx(2 * Q / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-3 + 0.0006 / line["ms_per_step"])
    txt = synth.text(2, n)
    oi = oracle.Index.from_text(txt.tobytes(), oracle.DNA, 8, seed_k)
    for rank in range(2):
        got = np.load(tmp_path / f"rank{rank}.npz")
        first = int(got["first"])
        assert first == rank * Q and int(got["queries"]) == Q
        q = (synth.planted_queries(103, Q, K, txt, first=first) if workload == "planted"
             else synth.random_queries(102, Q, K, first=first))
        chars, offsets = synth.fixed_csr(q)
        sp, ep, cnt, _ = oi.batch_search(chars, offsets, threads=8)
        if mode == "count":
            assert np.array_equal(got["counts"], cnt)
            assert cnt.max() > 0  # 13-mers against 3 Mbp: a good share of them occurs
            continue
        hit = cnt > 0
        assert hit.all()
        assert np.array_equal(got["ranges"][:, 0], sp) and np.array_equal(got["ranges"][:, 1], ep)
        ho, pos, _ = oi.batch_locate(sp, ep, threads=8)
        assert np.array_equal(got["hit_offsets"], ho) and np.array_equal(got["positions"], pos)
    # the two shards are different queries (rank 1 did not just repeat rank 0's)
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    key = "counts" if mode == "count" else "positions"
    assert a[key].shape != b[key].shape or not np.array_equal(a[key], b[key])


def _line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.parametrize("workload,mode,scaling", [("planted", "locate", "strong"), ("mixed", "count", "strong"),
                                                   ("random", "count", "weak")])
def test_n_rank_digests_equal_the_one_rank_run(require_gpu, tmp_path, workload, mode, scaling):
    """what the first real multi-GPU run is checked by: every rank's additive digest of its shard's counts / positions,
    summed on rank 0, must equal the digest the 1-rank run of the same batch recorded (strong: one batch cut in two,
    mixed lengths cut by the sum of the lengths; weak: the 2-rank batch is the 1-rank batch plus a second one)"""
    golden = str(tmp_path / "digests.json")
    common = ["--text-len", "3e6", "--queries", "1e6", "--kmer", "15", "--seed-k", "8", "--workload", workload, "--mode", mode,
              "--no-cpu", "--no-e2e", "--no-secondary", "--general-steps", "0", "--steps", "1", "--warmup", "1"]
    one = _line(_run_bench(common + ["--gpus", "1", "--scaling", scaling, "--record-digests", golden]))
    assert one["digests"]["status"] == "unknown"  # nothing committed for this toy configuration
    if scaling == "weak":  # the second rank's batch, recorded by a 1-rank run that starts at its first k-mer
        _line(_run_bench(common + ["--gpus", "1", "--query-offset", "1e6", "--record-digests", golden]))
    env_golden = dict(AWFM_BENCH_DIGESTS=golden)
    os.environ.update(env_golden)
    try:
        two = _line(_run_bench(common + ["--gpus", "2", "--force-device", "0", "--dist-backend", "gloo", "--scaling", scaling]))
        again = _line(_run_bench(common + ["--gpus", "1", "--scaling", scaling]))
    finally:
        os.environ.pop("AWFM_BENCH_DIGESTS")
    assert two["n_gpus"] == 2 and two["scaling"] == scaling
    assert two["digests"]["status"] == "match" and two["digests"]["shards"] == 2, two["digests"]
    assert again["digests"]["status"] == "match"
    if scaling == "strong":
        assert two["config"]["batch_kmers"] == 1_000_000 and two["digests"]["counts"] == one["digests"]["counts"]
        firsts = [p["first"] for p in two["digests"]["per_rank"]]
        assert firsts[0] == 0 and 0 < firsts[1] < 1_000_000
        if workload == "mixed":
            assert firsts[1] != 500_000  # cut by the sum of the lengths, not by the number of k-mers
    else:
        assert two["config"]["batch_kmers"] == 2_000_000


@pytest.mark.gpu
def test_seed_bucket_sharding_digests_equal_the_contiguous_one_rank_run(require_gpu, tmp_path):
    """round 6: --sharding seed_bucket -- every rank orders its stretch of the batch, the ranks exchange the records by bucket
    range (here: two ranks on one GPU, the all-to-all through host memory over gloo), every rank searches a dense half of the
    seed order.  The results carry the k-mers' numbers in the whole batch, so the two ranks' keyed digests must add up to the
    digest the ordinary 1-rank run of the same batch recorded; and so must the 1-rank run of the new mode itself."""
    golden = str(tmp_path / "digests.json")
    common = ["--text-len", "3e6", "--queries", "1e6", "--kmer", "21", "--seed-k", "8", "--workload", "planted", "--mode", "locate",
              "--no-cpu", "--no-e2e", "--no-secondary", "--general-steps", "0", "--steps", "1", "--warmup", "1", "--scaling", "strong"]
    one = _line(_run_bench(common + ["--gpus", "1", "--record-digests", golden]))
    os.environ["AWFM_BENCH_DIGESTS"] = golden
    try:
        # (two batches in flight, the default: three timed steps so that both slots of the pipeline are used again; and one
        # batch at a time)
        two = _line(_run_bench(common + ["--gpus", "2", "--force-device", "0", "--dist-backend", "gloo", "--sharding", "seed_bucket", "--steps", "3"]))
        alone = _line(_run_bench(common + ["--gpus", "1", "--sharding", "seed_bucket", "--seed-bucket-pipeline", "0"]))
    finally:
        os.environ.pop("AWFM_BENCH_DIGESTS")
    for line in (two, alone):
        assert line["config"]["sharding"] == "seed_bucket" and line["digests"]["status"] == "match", line["digests"]
        assert line["digests"]["counts"] == one["digests"]["counts"] and line["digests"]["positions"] == one["digests"]["positions"]
    assert two["config"]["batches_in_flight"] == 2 and alone["config"]["batches_in_flight"] == 1
    assert two["n_gpus"] == 2 and len(two["digests"]["per_rank"]) == 2
    kmers = [p["kmers"] for p in two["digests"]["per_rank"]]
    assert sum(kmers) == 1_000_000 and min(kmers) > 300_000  # two dense halves of the order, not two halves of the batch


@pytest.mark.gpu
def test_two_ranks_on_two_distinct_gpus_over_rccl(require_gpu, tmp_path):
    """--gpus 2 the way the driver runs it at N > 1: one rank per device, barrier and MAX reduction over RCCL"""
    from avxwindowfmindex_amd import _lib
    if _lib.lib().awfmGpuDeviceCount() < 2:
        pytest.skip("needs two GPUs")
    golden = str(tmp_path / "digests.json")
    common = ["--text-len", "3e6", "--queries", "1e6", "--kmer", "15", "--seed-k", "8", "--workload", "planted", "--no-cpu",
              "--no-e2e", "--no-secondary", "--general-steps", "0", "--steps", "2", "--warmup", "1", "--scaling", "strong"]
    _line(_run_bench(common + ["--gpus", "1", "--record-digests", golden]))
    os.environ["AWFM_BENCH_DIGESTS"] = golden
    try:
        two = _line(_run_bench(common + ["--gpus", "2"]))
    finally:
        os.environ.pop("AWFM_BENCH_DIGESTS")
    assert two["n_gpus"] == 2 and two["digests"]["status"] == "match"
    assert two["config"]["timing_collective"] in ("nccl", "gloo")


@pytest.mark.gpu
def test_aos_locate_shards_over_two_devices(oracle, awfm, require_gpu):
    """AWFM_GPU_DEVICES=0,1: awFmParallelSearchLocate cuts the list over two devices, one index replica each"""
    from avxwindowfmindex_amd import _lib, synth
    if _lib.lib().awfmGpuDeviceCount() < 2:
        pytest.skip("needs two GPUs")
    os.environ["AWFM_GPU_DEVICES"] = "0,1"
    try:
        txt = synth.text(91, 400_000)
        ix = awfm.create_index(txt, awfm.AwFmAlphabetDna, 8, 8)
        kmers = np.concatenate([synth.random_queries(92, 20_000, 19), synth.planted_queries(93, 20_001, 19, txt)])
        lst = awfm.KmerSearchList(len(kmers))
        lst.fill(kmers)
        awfm.parallel_search_locate(ix, lst, 8)
        oi = oracle.Index.wrap(oracle.DNA, 8, 8, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
        chars, offsets = synth.fixed_csr(kmers)
        sp, ep, cnt, _ = oi.batch_search(chars, offsets)
        ho, pos, _ = oi.batch_locate(sp, ep)
        assert np.array_equal(lst.counts(), cnt)
        for i in list(range(0, len(kmers), 997)) + [len(kmers) - 1]:
            assert np.array_equal(lst.positions(i), pos[ho[i]:ho[i + 1]])
        lst.dealloc()
        ix.dealloc()
    finally:
        os.environ.pop("AWFM_GPU_DEVICES")
