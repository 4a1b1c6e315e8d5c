#!/usr/bin/env python3
"""Where the first construction of the deeper table spends its time (AWFM_VERBOSE timings of every level, then the next-step
bits): scripts/deep_build_probe.py [text length]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AWFM_VERBOSE"] = "1"
import torch  # noqa: E402
from avxwindowfmindex_amd import _lib, api  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
L = _lib.lib()
d_text = torch.empty(n, dtype=torch.uint8, device="cuda")
L.awfmGpuSynthText(d_text.data_ptr(), 0, n, 2, 0, None)
os.environ["AWFM_GPU_DEEP_SEED_K"] = "0"
os.environ["AWFM_GPU_DENSE_SA"] = "0"
t0 = time.time()
ix = api.gpu_create_index(d_text.data_ptr(), api.AwFmAlphabetDna, 8, 12, on_device_length=n)
g = api.GpuIndex(ix, acquire=True)
torch.cuda.synchronize()
print(f"index + image without the accelerators: {time.time() - t0:.2f} s", flush=True)
for attempt in ("first", "second"):
    t0 = time.time()
    g.set_deep_seed(16)
    torch.cuda.synchronize()
    print(f"{attempt} construction of the depth-16 table: {time.time() - t0:.2f} s (library: {g.deep_seed_build})", flush=True)
    g.set_deep_seed(0)
t0 = time.time()
x = torch.empty(34 * (1 << 30), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
print(f"torch.empty of 34 GiB: {time.time() - t0:.2f} s", flush=True)
