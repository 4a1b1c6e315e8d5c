#!/usr/bin/env python3
"""PCIe- and malloc-inclusive rate of the drop-in AoS API (awFmParallelSearchCount / Locate) for DESIGN.md."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avxwindowfmindex_amd import _lib, api, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 32
K = 21
import torch  # noqa: E402
L = _lib.lib()
d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 2, 0, None)
ix = api.gpu_create_index(d_text.data_ptr(), api.AwFmAlphabetDna, 8, 12, on_device_length=n)
d_q = torch.empty(Q * K, dtype=torch.uint8, device="cuda")
L.awfmGpuSynthPlantedQueries(d_q.data_ptr(), 0, Q, K, 103, d_text.data_ptr(), n, None)
q = np.ascontiguousarray(d_q.cpu().numpy())
lst = api.KmerSearchList(Q)
data = lst.ptr.contents.kmerSearchData
base = q.ctypes.data
t0 = time.time()
arr = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_uint64)), shape=(Q, 4))
arr[:, 0] = base + np.arange(Q, dtype=np.uint64) * np.uint64(K)
arr[:, 1] = K
lst.ptr.contents.count = Q
print(f"list fill {time.time() - t0:.2f}s")
for name, fn in (("count", lambda: api.parallel_search_count(ix, lst, threads)),
                 ("locate", lambda: api.parallel_search_locate(ix, lst, threads))):
    fn()
    t0 = time.time()
    fn()
    dt = time.time() - t0
    print(f"awFmParallelSearch{name.capitalize()} AoS, {Q} planted {K}-mers, {threads} host threads: {dt*1e3:.1f} ms = {Q/dt/1e6:.1f} Mkmers/s")
