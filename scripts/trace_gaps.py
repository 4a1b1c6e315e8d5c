#!/usr/bin/env python3
"""Timeline of kernels of a rocprofv3 --kernel-trace run: name, duration, gap since the previous kernel ended.
usage: trace_gaps.py <dir with *_kernel_trace.csv> [last N kernels, default 80]
       trace_gaps.py <dir> --around <kernel name part> [occurrences, default 3]   (the launches around the last few of them)"""
import csv
import glob
import os
import sys

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if len(sys.argv) > 2 and sys.argv[2] == "--around":
    pat = sys.argv[3]
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    idx = [i for i, r in enumerate(rows) if pat in r[2]]
    if not idx:
        sys.exit(f"no kernel named *{pat}*")
    rows = rows[max(idx[-k:][0] - 8, 0): idx[-1] + 12]
else:
    rows = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 80):]
t0 = rows[0][0]
prev_end = rows[0][0]
busy = 0
for s, e, name in rows:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {short}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us")
