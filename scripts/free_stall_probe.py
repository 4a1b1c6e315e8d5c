#!/usr/bin/env python3
"""Who pays for freed device memory?  hipMalloc N GB, touch it, hipFree it, then time the next allocations: the first hipMalloc
after a large hipFree carries the seconds (round 5: first_call_s 7.0 on the driver's box).  usage: free_stall_probe.py [GB ...]"""
import ctypes as C
import sys
import time

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]


def malloc(nbytes):
    p = C.c_void_p()
    t0 = time.perf_counter()
    rc = hip.hipMalloc(C.byref(p), nbytes)
    return p, time.perf_counter() - t0, rc


def timed(fn, *a):
    t0 = time.perf_counter()
    rc = fn(*a)
    return time.perf_counter() - t0, rc


assert hip.hipSetDevice(0) == 0
p, dt, rc = malloc(1 << 20)
print(f"first hipMalloc of the process (1 MB): {dt:.3f} s rc {rc}")
hip.hipFree(p)
for gb in [float(x) for x in sys.argv[1:]] or [8, 40, 80]:
    n = int(gb * (1 << 30))
    p, dt_alloc, rc = malloc(n)
    dt_set, _ = timed(hip.hipMemset, p, 1, n)
    dt_sync, _ = timed(hip.hipDeviceSynchronize)
    dt_free, _ = timed(hip.hipFree, p)
    q, dt_next, _ = malloc(4 << 30)
    dt_set2, _ = timed(hip.hipMemset, q, 1, 4 << 30)
    dt_sync2, _ = timed(hip.hipDeviceSynchronize)
    hip.hipFree(q)
    r, dt_third, _ = malloc(4 << 30)
    hip.hipFree(r)
    print(f"{gb:5.0f} GB: hipMalloc {dt_alloc:.3f} s, memset+sync {dt_set + dt_sync:.3f} s, hipFree {dt_free:.3f} s | next hipMalloc(4 GB) {dt_next:.3f} s, "
          f"its memset+sync {dt_set2 + dt_sync2:.3f} s | the one after {dt_third:.3f} s", flush=True)
# the same again: does a second round of the same sizes cost the same?
for gb in [40]:
    n = int(gb * (1 << 30))
    p, dt_alloc, rc = malloc(n)
    hip.hipMemset(p, 1, n)
    hip.hipDeviceSynchronize()
    dt_free, _ = timed(hip.hipFree, p)
    dt_sync, _ = timed(hip.hipDeviceSynchronize)
    q, dt_small, _ = malloc(1 << 20)
    hip.hipFree(q)
    r, dt_next, _ = malloc(34 << 30)
    hip.hipFree(r)
    print(f"again {gb} GB: hipMalloc {dt_alloc:.3f} s, hipFree {dt_free:.3f} s, sync {dt_sync:.3f} s, then hipMalloc(1 MB) {dt_small:.3f} s, hipMalloc(34 GB) {dt_next:.3f} s")
# near the device's capacity, in pieces (what the index builder's sort does): 8 x 26 GB, used, freed, then the image's allocations
ps = []
for i in range(8):
    p, dt, rc = malloc(26 << 30)
    assert rc == 0, rc
    hip.hipMemset(p, 1, 26 << 30)
    ps.append(p)
hip.hipDeviceSynchronize()
t0 = time.perf_counter()
for p in ps:
    hip.hipFree(p)
dt_free = time.perf_counter() - t0
q, dt1, _ = malloc(4 << 30)
r, dt2, _ = malloc(34 << 30)
dt_set, _ = timed(hip.hipMemset, r, 1, 34 << 30)
dt_sync, _ = timed(hip.hipDeviceSynchronize)
print(f"8 x 26 GB used and freed ({dt_free:.3f} s): hipMalloc(4 GB) {dt1:.3f} s, hipMalloc(34 GB) {dt2:.3f} s, first touch of it {dt_set + dt_sync:.3f} s")
