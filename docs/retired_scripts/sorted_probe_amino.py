#!/usr/bin/env python3
"""Probe: the amino search kernel on a batch ordered by seed (Swiss-Prot-sized synthetic index, 50 M 10-mers, k=5).

MI355X, 128-B amino blocks: unsorted 3.53 ms, sorted by seed 2.61 ms (round 1, 256-B blocks: 5.28 / 4.04 ms) -- the
0.9 ms are what encoding and sorting 5*10^7 k-mers costs (0.3 + 0.7 ms), so the amino alphabet has no ordered path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AWFM_GPU_BLOCKS_PER_CU", "8")
import torch  # noqa: E402
from avxwindowfmindex_amd import _lib, api, synth  # noqa: E402

n, K, SEEDK = 200_000_000, 10, 5
NG = 131072
Q = NG * 381
L = _lib.lib()
dev = torch.device("cuda")
d_text = torch.empty(n, dtype=torch.uint8, device=dev)
L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 4, 1, None)
ix = api.gpu_create_index(d_text.data_ptr(), api.AwFmAlphabetAmino, 8, SEEDK, on_device_length=n)
del d_text
g = api.GpuIndex(ix, acquire=True)
d_q = torch.empty(Q * K, dtype=torch.uint8, device=dev)
L.awfmGpuSynthRandomQueries(d_q.data_ptr(), 0, Q, K, 104, 1, None)
q2 = d_q.view(Q, K)
lut = torch.zeros(256, dtype=torch.int64, device=dev)
for i, c in enumerate(synth.AMINO_ALPHABET):
    lut[c] = i
    lut[c & 0xDF] = i
key = torch.zeros(Q, dtype=torch.int64, device=dev)
for j in range(K - SEEDK, K):
    key = key * 20 + lut[q2[:, j].long()]
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
    print(f"{name:40s} {min(ts[1:]):7.2f} ms  {Q / min(ts[1:]) / 1e3:8.0f} Mkmers/s", flush=True)


def xcd_contiguous(order):
    p = torch.arange(Q, device=dev)
    it, rem = p // NG, p % NG
    b, r = rem // 64, rem % 64
    x, l = b % 8, b // 8
    return order[x * (Q // 8) + (it * 256 + l) * 64 + r]


run("unsorted", d_q)
full = torch.argsort(key)
run("sorted by seed", q2[full].contiguous().view(-1))
run("sorted by seed, XCD-contiguous", q2[xcd_contiguous(full)].contiguous().view(-1))
