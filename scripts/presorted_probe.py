#!/usr/bin/env python3
"""Probe: what would a finer order be worth to the ordered search?  The batch is pre-sorted by MORE bits of the seed than
the 15 the sort key has (the radix sort is stable, so the finer order survives inside every key bucket), and
orderedSearchKernel's own time is read from its HIP events (GRCh38-sized index, 10^8 random / planted 21-mers, counts).
usage: scripts/presorted_probe.py [text length] [k-mers]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AWFM_GPU_TIME_ORDERED"] = "1"
import torch  # noqa: E402
from avxwindowfmindex_amd import _lib, api  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
Q = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
K, SEEDK = 21, 12
L = _lib.lib()
dev = torch.device("cuda")
d_text = torch.empty(n, dtype=torch.uint8, device=dev)
L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 2, 0, None)
ix = api.gpu_create_index(d_text.data_ptr(), api.AwFmAlphabetDna, 8, SEEDK, on_device_length=n)
g = api.GpuIndex(ix, acquire=True)
g.set_ordered(1)
lut = torch.zeros(256, dtype=torch.int64, device=dev)
for i, c in enumerate(b"acgt"):
    lut[c] = i
d_counts = torch.empty(Q, dtype=torch.int32, device=dev)


def measure(name, chars):
    ts, ks = [], []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.search_hits(chars.data_ptr(), 0, K, Q, 0, d_counts.data_ptr())
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
        ks.append(g.last_ordered_kernel_ms())
    print(f"{name:44s} call {min(ts[1:]):6.2f} ms   orderedSearchKernel {min(ks[1:]):6.2f} ms   hits {int((d_counts > 0).sum())}", flush=True)


DEPTH = g.deep_seed_k or SEEDK  # the table the search starts from
for workload in ("random", "planted"):
    d_q = torch.empty(Q * K, dtype=torch.uint8, device=dev)
    if workload == "random":
        L.awfmGpuSynthRandomQueries(d_q.data_ptr(), 0, Q, K, 102, 0, None)
    else:
        L.awfmGpuSynthPlantedQueries(d_q.data_ptr(), 0, Q, K, 103, d_text.data_ptr(), n, None)
    q2 = d_q.view(Q, K)
    measure(f"{workload}: as generated", d_q)
    # key over the characters the search consumes first: the table index (last DEPTH characters, first one most
    # significant), then the ones before it; the partition pass scrambles inside a tile of 16384 k-mers only, so a sorted
    # batch reaches the search kernel in an order that fine
    for extra in (0, 2):
        key = torch.zeros(Q, dtype=torch.int64, device=dev)
        for j in range(K - DEPTH, K):
            key = key * 4 + lut[q2[:, j].long()]
        for j in range(K - DEPTH - 1, K - DEPTH - 1 - extra, -1):
            key = key * 4 + lut[q2[:, j].long()]
        order = torch.argsort(key)
        del key
        sorted_q = q2[order].contiguous().view(-1)
        del order
        measure(f"{workload}: pre-sorted by the table index + {extra} characters", sorted_q)
        del sorted_q
    del d_q, q2
