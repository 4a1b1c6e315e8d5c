#!/usr/bin/env python3
"""Probe: how much faster is the search kernel when the batch is ordered/bucketed by seed (L2 locality)?

MI355X, GRCh38-sized image, 99.9 M random 21-mers (searchKernel G=4): unsorted 14.3 ms, sorted by the 24-bit
seed 7.8 ms, sorted by its top 16 bits 11.3 ms (8.2 ms when every XCD gets a contiguous eighth of the order),
top 12 bits 12.1 / 11.1 ms, top 8 bits 13.1 / 12.4 ms.  See DESIGN.md 4c for why the product does not do this."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AWFM_GPU_BLOCKS_PER_CU", "8")
import torch  # noqa: E402
from avxwindowfmindex_amd import _lib, api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
iters = 762
NG = 131072
Q = NG * iters
K, SEEDK = 21, 12
L = _lib.lib()
dev = torch.device("cuda")
d_text = torch.empty(n, dtype=torch.uint8, device=dev)
L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 2, 0, None)
ix = api.gpu_create_index(d_text.data_ptr(), api.AwFmAlphabetDna, 8, SEEDK, on_device_length=n)
del d_text
g = api.GpuIndex(ix, acquire=True)
d_q = torch.empty(Q * K, dtype=torch.uint8, device=dev)
L.awfmGpuSynthRandomQueries(d_q.data_ptr(), 0, Q, K, 101, 0, None)
q2 = d_q.view(Q, K)
lut = torch.zeros(256, dtype=torch.int64, device=dev)
for i, c in enumerate(b"ACGT"):
    lut[c] = i
    lut[c | 0x20] = i
    lut[c | 0x20] = i
key = torch.zeros(Q, dtype=torch.int64, device=dev)
for j in range(K - SEEDK, K):
    key = key * 4 + lut[q2[:, j].long()]
assert int(key.max()) > 0
assert int(key.max()) > 0
d_ranges = torch.empty(Q * 2, dtype=torch.int64, device=dev)


def run(name, chars):
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.search(chars.data_ptr(), 0, K, Q, d_ranges.data_ptr(), 0)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"{name:44s} {min(ts[1:]):7.2f} ms  {Q / min(ts[1:]) / 1e3:8.0f} Mkmers/s", flush=True)


def xcd_contiguous(order):
    """re-arrange so that XCD x (workgroups b with b%8==x) walks the x-th eighth of `order` front to back"""
    p = torch.arange(Q, device=dev)
    it, rem = p // NG, p % NG
    b, r = rem // 64, rem % 64
    x, l = b % 8, b // 8
    src = x * (Q // 8) + (it * 256 + l) * 64 + r
    return order[src]


run("unsorted", d_q)
full = torch.argsort(key)
run("sorted by seed", q2[full].contiguous().view(-1))
if os.environ.get("PROBE_SHORT"):
    sys.exit(0)
run("sorted by seed, XCD-contiguous", q2[xcd_contiguous(full)].contiguous().view(-1))
for bits in (16, 12, 8):
    o = torch.argsort(key >> (24 - bits), stable=True)
    run(f"bucketed by top {bits} bits", q2[o].contiguous().view(-1))
    run(f"bucketed by top {bits} bits, XCD-contiguous", q2[xcd_contiguous(o)].contiguous().view(-1))
