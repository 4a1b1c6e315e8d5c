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
                    "--steps", "2", "--warmup", "1", "--dump-dir", str(tmp_path)])
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
