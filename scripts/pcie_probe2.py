#!/usr/bin/env python3
"""which engine moves device-to-host copies, by API (rocprofv3 --kernel-trace --memory-copy-trace tells)"""
import ctypes as C, os, sys, time
import torch
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemcpyDtoHAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
hip.hipMemcpyWithStream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
n = 128 << 20
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
host_t = torch.empty(n, dtype=torch.uint8).pin_memory()
s = torch.cuda.Stream()
torch.cuda.synchronize()
def timed(label, fn):
    for _ in range(2):
        t0 = time.perf_counter(); rc = fn(); s.synchronize()
    print(f"{label} rc {rc} {n / (time.perf_counter() - t0) / 1e9:.1f} GB/s", flush=True); time.sleep(0.05)
timed("A hipMemcpyAsync kind=D2H", lambda: hip.hipMemcpyAsync(host_t.data_ptr(), dev.data_ptr(), n, 2, s.cuda_stream))
timed("B hipMemcpyAsync kind=Default", lambda: hip.hipMemcpyAsync(host_t.data_ptr(), dev.data_ptr(), n, 4, s.cuda_stream))
timed("C hipMemcpyDtoHAsync", lambda: hip.hipMemcpyDtoHAsync(host_t.data_ptr(), dev.data_ptr(), n, s.cuda_stream))
timed("D hipMemcpyWithStream", lambda: hip.hipMemcpyWithStream(host_t.data_ptr(), dev.data_ptr(), n, 2, s.cuda_stream))
def torch_copy():
    with torch.cuda.stream(s): host_t.copy_(dev, non_blocking=True)
    return 0
timed("E torch copy_ non_blocking", torch_copy)
def torch_copy_default():
    host_t.copy_(dev, non_blocking=True); torch.cuda.synchronize(); return 0
timed("F torch copy_ default stream", torch_copy_default)
timed("G null stream hipMemcpyAsync", lambda: hip.hipMemcpyAsync(host_t.data_ptr(), dev.data_ptr(), n, 2, None))
