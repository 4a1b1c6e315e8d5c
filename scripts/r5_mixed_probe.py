#!/usr/bin/env python3
"""cfg 5's batch (10^8 k-mers of 8..30 characters, counted) through awfmGpuSearchHits, timed call by call with knobs given on
the command line as ENV=VALUE words (each variant is the default + one word): where mixedLookupSearchKernel's time goes.
usage: scripts/r5_mixed_probe.py [ENV=VALUE ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from avxwindowfmindex_amd import _lib, api  # noqa: E402

n, Q = 3_100_000_000, 100_000_000
L = _lib.lib()
dev = torch.device("cuda")
d_text = torch.empty(n, dtype=torch.uint8, device=dev)
assert L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 2, 0, None) == 1
ix = api.gpu_create_index(d_text.data_ptr(), api.AwFmAlphabetDna, 8, 12, on_device_length=n)
g = api.GpuIndex(ix, acquire=True)
lens = torch.empty(Q, dtype=torch.int64, device=dev)
assert L.awfmGpuSynthMixedLengths(lens.data_ptr(), 0, Q, 8, 30, 105, None) == 1
off = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
torch.cumsum(lens, 0, out=off[1:])
chars = torch.empty(int(off[-1].item()) + 64, dtype=torch.uint8, device=dev)
assert L.awfmGpuSynthMixedQueries(chars.data_ptr(), off.data_ptr(), 0, Q, 105, d_text.data_ptr(), n, 0, None) == 1
del d_text, lens
counts = torch.empty(Q, dtype=torch.int32, device=dev)
stream = torch.cuda.Stream()
os.environ["AWFM_GPU_TIME_ORDERED"] = "1"


def run(word):
    if word:
        k, v = word.split("=", 1)
        os.environ[k] = v
    for _ in range(2):
        g.search_hits(chars.data_ptr(), off.data_ptr(), 0, Q, 0, counts.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    g.ordered_kernel_log()
    t0 = time.perf_counter()
    for _ in range(5):
        g.search_hits(chars.data_ptr(), off.data_ptr(), 0, Q, 0, counts.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 5
    log = g.ordered_kernel_log()
    kms = sum(f for f, _ in log) / len(log)
    print(f"{word or 'default':40s} call {ms:7.3f} ms   lookup kernel {kms:7.3f} ms   k-mers with hits {int((counts != 0).sum().item())}", flush=True)
    if word:
        del os.environ[word.split("=", 1)[0]]


run("")
for w in sys.argv[1:]:
    run(w)
run("")
