#!/usr/bin/env python3
"""the packed pipeline alone (for rocprofv3 --kernel-trace --stats): GRCh38-sized index, 10^8 planted 21-mers, locate"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from avxwindowfmindex_amd import _lib, api  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
Q = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
workload = sys.argv[3] if len(sys.argv) > 3 else "planted"
chunk = int(float(sys.argv[4])) if len(sys.argv) > 4 else 0
K = 21
L = _lib.lib()
d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 2, 0, None)
ix = api.gpu_create_index(d_text.data_ptr(), api.AwFmAlphabetDna, 8, 12, on_device_length=n)
g = api.GpuIndex(ix, acquire=True)
d_q = torch.empty(Q * K, dtype=torch.uint8, device="cuda")
if workload == "planted":
    L.awfmGpuSynthPlantedQueries(d_q.data_ptr(), 0, Q, K, 103, d_text.data_ptr(), n, None)
else:
    L.awfmGpuSynthRandomQueries(d_q.data_ptr(), 0, Q, K, 102, 0, None)
del d_text
d_packed = torch.empty(Q, dtype=torch.int64, device="cuda")
assert g.pack_device(d_q.data_ptr(), K, Q, d_packed.data_ptr()) == 0
address = L.awfmGpuHostAlloc(Q * 8)
np.ctypeslib.as_array(C.cast(address, C.POINTER(C.c_uint64)), shape=(Q,))[:] = d_packed.cpu().numpy().view(np.uint64)
del d_packed, d_q
torch.cuda.empty_cache()
seen = {"hits": 0}


def sink(user, first, m, counts, positions, total):
    seen["hits"] += total
    return 0


for i in range(4):
    seen["hits"] = 0
    t0 = time.perf_counter()
    g.stream((address, Q), K, locate=True, chunk=chunk, sink=sink)
    dt = time.perf_counter() - t0
    print(f"run {i}: {dt * 1e3:.1f} ms = {Q / dt / 1e6:.0f} Mkmers/s, {seen['hits']} hits", flush=True)
