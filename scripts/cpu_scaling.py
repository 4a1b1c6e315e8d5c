#!/usr/bin/env python3
"""Thread scaling of the CPU oracle (the cpu_baseline port) on the GRCh38-sized index."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avxwindowfmindex_amd import _lib, api
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
L = _lib.lib()
d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 2, 0, None)
ix = api.gpu_create_index(d_text.data_ptr(), api.AwFmAlphabetDna, 8, 12, on_device_length=n)
Q, K = 20_000_000, 21
d_q = torch.empty(Q * K, dtype=torch.uint8, device="cuda")
L.awfmGpuSynthRandomQueries(d_q.data_ptr(), 0, Q, K, 102, 0, None)
chars = d_q.cpu().numpy()
offsets = np.arange(Q + 1, dtype=np.uint64) * np.uint64(K)
oi = O.Index.wrap(O.DNA, 8, 12, ix.bwt_length, ix.blocks(), ix.prefix_sums(), ix.seed_table(), ix.packed_sa())
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cpu.max n/a", e)
for threads in (1, 4, 16, 64, 128, 256):
    m = min(Q, 400_000 * threads)
    t0 = time.perf_counter()
    oi.batch_search(chars[: m * K], offsets[: m + 1], threads=threads)
    dt = time.perf_counter() - t0
    print(f"threads {threads:4d}: {m / dt / 1e6:8.2f} Mkmers/s  ({m} queries, {dt:.2f} s)", flush=True)
