#!/usr/bin/env python3
"""PCIe rates seen by hipMemcpyAsync between page-locked host memory and the device: one stream, then two at once."""
import time
import torch
n = 256 << 20
host = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
dev = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
def run(label, fn, total):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    print(f"{label:28s} {total / dt / 1e9:7.1f} GB/s")
def h2d(i):
    with torch.cuda.stream(streams[i]): dev[i].copy_(host[i], non_blocking=True)
def d2h(i):
    with torch.cuda.stream(streams[i]): host[i].copy_(dev[i], non_blocking=True)
run("H2D one stream", lambda: h2d(0), n)
run("D2H one stream", lambda: d2h(0), n)
run("H2D two streams", lambda: (h2d(0), h2d(1)), 2 * n)
run("D2H two streams", lambda: (d2h(0), d2h(1)), 2 * n)
run("H2D + D2H at once", lambda: (h2d(0), d2h(1)), 2 * n)
