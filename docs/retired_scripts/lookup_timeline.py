#!/usr/bin/env python3
"""Where a lookupSearchKernel launch spends its time beside its trips: the waves' start and end times
($AWFM_GPU_LOOKUP_TIMELINE=<file>, the device's 100-MHz clock), where they ran (HW_ID) and how many trips they made, of the
last launch of a bench run.
usage: AWFM_GPU_LOOKUP_TIMELINE=/tmp/t.bin python bench.py ... ; scripts/lookup_timeline.py /tmp/t.bin"""
import sys

import numpy as np

t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
wg = np.arange(len(t)) // 4
keep = (t[:, 0] != 0) & (t[:, 1] != 0)
t, wg = t[keep].astype(np.int64), wg[keep]
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0  # microseconds
hw, trips = t[:, 2], t[:, 3]
# gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx94x: 3 bits), ...
wave_id, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
print(f"{len(t)} waves; kernel span {end.max():.1f} us; busy per wave mean {np.mean(end - start):.1f} us (min {np.min(end - start):.1f}, max {np.max(end - start):.1f}); trips per wave {trips.min()}..{trips.max()} (mean {trips.mean():.1f})")
print("start  us: " + "  ".join(f"p{p}={np.percentile(start, p):.1f}" for p in (0, 10, 50, 90, 99, 100)))
print("end    us: " + "  ".join(f"p{p}={np.percentile(end, p):.1f}" for p in (0, 1, 10, 50, 90, 100)))
idle = (start.sum() + (end.max() - end).sum()) / (len(t) * end.max())
print(f"share of the wave-time of the span spent before a wave's start or after its end: {idle:.3f} "
      f"(before: {start.mean():.1f} us a wave, after: {(end.max() - end).mean():.1f} us a wave)")
per_trip = (end - start) / np.maximum(trips, 1)
def by(name, key):
    print(f"  by {name}: " + "  ".join(f"{k}: end {np.median(end[key == k]):.0f} us/trip {np.median(per_trip[key == k]):.1f} n={int((key == k).sum())}" for k in np.unique(key)))
by("XCD (workgroup % 8)", wg % 8)
by("wave slot", wave_id)
by("SIMD", simd)
by("SE", se)
by("CU in its SH", cu)
