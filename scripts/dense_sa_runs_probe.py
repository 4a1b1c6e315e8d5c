#!/usr/bin/env python3
"""The full suffix array of the genome-shaped 3.1 Gbp text (24 runs of N of 10^5..10^7 characters), two ways: the array the
GPU builder hands to the image of its index, and the AUTOMATIC construction a loaded index gets (walks capped at 32 x ratio
steps, the parked ones completed by pointer jumping).  Same positions for 10^6 k-mers drawn from the text, the k-mers right
behind every run (their LF walks enter it) and a k-mer of 32 N (2.4 * 10^8 hits, all of them inside the runs)?
usage: scripts/dense_sa_runs_probe.py [text length, default 3.1e9]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from avxwindowfmindex_amd import _lib, api  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
L = _lib.lib()
dev = torch.device("cuda")
d_text = torch.empty(n, dtype=torch.uint8, device=dev)
assert L.awfmGpuSynthGenomeText(d_text.data_ptr(), n, 101, None) == 1
t0 = time.perf_counter()
ix = api.gpu_create_index(d_text.data_ptr(), api.AwFmAlphabetDna, 8, 12, on_device_length=n)
g1 = api.GpuIndex(ix, acquire=True)
print(f"built + image in {time.perf_counter() - t0:.1f} s; full suffix array from the builder: {g1.has_dense_sa} ({g1.dense_sa_build_s:.3f} s)", flush=True)
# k-mers: planted clean ones, the 16-mers right behind every run of N, 32 N
Q, K = 1_000_000, 21
d_q = torch.empty(Q * K, dtype=torch.uint8, device=dev)
assert L.awfmGpuSynthPlantedQueriesClean(d_q.data_ptr(), 0, Q, K, 7, d_text.data_ptr(), n, None) == 1
ends = []  # first character behind a run (in pieces: torch.nonzero indexes with 32 bits)
piece = 1 << 30
for b in range(0, n - 1, piece):
    e = min(n - 1, b + piece)
    is_n = d_text[b:e + 1] == ord("n")
    ends += (torch.nonzero(is_n[:-1] & ~is_n[1:]).flatten() + (b + 1)).tolist()
    del is_n
extra = [d_text[int(e): int(e) + K].cpu().numpy().tobytes() for e in ends if int(e) + K <= n]
extra = [k for k in extra if b"n" not in k]
print(f"{len(extra)} runs of N", flush=True)
chars = np.concatenate([d_q.cpu().numpy(), np.frombuffer(b"".join(extra), np.uint8), np.frombuffer(b"n" * K, np.uint8)])
total = Q + len(extra) + 1
d_chars = torch.from_numpy(chars).to(dev)
del d_text, d_q
torch.cuda.empty_cache()


def locate(g):
    d_ranges = torch.empty(total * 2, dtype=torch.int64, device=dev)
    g.search(d_chars.data_ptr(), 0, K, total, d_ranges.data_ptr(), 0)
    d_off = torch.empty(total + 1, dtype=torch.int64, device=dev)
    d_scratch = torch.empty(api.GpuIndex.scan_scratch_bytes(total), dtype=torch.uint8, device=dev)
    hits = g.hit_offsets(d_ranges.data_ptr(), total, d_off.data_ptr(), d_scratch.data_ptr())
    d_pos = torch.empty(max(hits, 1), dtype=torch.int64, device=dev)
    g.locate(d_ranges.data_ptr(), d_off.data_ptr(), total, hits, d_pos.data_ptr())
    torch.cuda.synchronize()
    return d_off, d_pos


off1, pos1 = locate(g1)
print(f"{int(off1[-1])} hits ({int(off1[-1] - off1[-2])} of them of the k-mer of N)", flush=True)
os.environ["AWFM_GPU_DENSE_SA"] = "auto"
os.environ["AWFM_VERBOSE"] = "1"
t0 = time.perf_counter()
g2 = api.GpuIndex(ix)  # an image of its own from the host arrays: what a loaded index gets
print(f"second image in {time.perf_counter() - t0:.1f} s; automatic full suffix array: {g2.has_dense_sa} ({g2.dense_sa_build_s:.2f} s)", flush=True)
assert g2.has_dense_sa
off2, pos2 = locate(g2)
assert torch.equal(off1, off2) and torch.equal(pos1, pos2), "the two full suffix arrays give different positions"
print("same positions", flush=True)
