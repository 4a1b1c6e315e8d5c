#!/usr/bin/env python3
"""The figures DESIGN.md / README.md / profiles/<round>/README.md quote, straight from profiles/<round>/: one line per bench
configuration and one per counter set.  usage: scripts/doc_numbers.py [profiles/r5]"""
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "profiles/r5"
print("== bench lines: value Mkmers/s, ms/step, kernel, kernel_ms, frac, hbm_frac_measured, reference_algorithm_frac, proxy N=8 (ms, efficiency)")
for f in sorted(glob.glob(os.path.join(root, "bench_*.json"))):
    if "under_rocprofv3" in f:
        continue
    try:
        d = json.load(open(f))
    except Exception as e:  # noqa: BLE001
        print(os.path.basename(f), "unreadable", e)
        continue
    r = d.get("roofline") or {}
    sp = (d.get("scaling_proxy") or {}).get("shards", {})
    e8 = {k: (v.get("8", {}).get("ms_max"), v.get("8", {}).get("efficiency")) for k, v in sp.items()}
    c = d.get("cpu_baseline") or {}
    print(f"{os.path.basename(f)[6:-5]:24s} {d['value']:9.1f} {d['ms_per_step']:7.3f} {str(r.get('kernel'))[:24]:24s} {r.get('kernel_ms')} frac {r.get('frac')} "
          f"hbm {r.get('hbm_frac_measured')} ref {r.get('reference_algorithm_frac')} cpu {c.get('value')}/{c.get('cores')} {e8 if e8 else ''}")
print("== counter sets: kernel, ms in the kernel trace, GB read + written per launch, TB/s, L2 hit rate, waves waiting, bench kernel_ms, commit")
for f in sorted(glob.glob(os.path.join(root, "counters_*.json"))):
    d = json.load(open(f))
    ns, rd, wr = d.get("avg_ns_kernel_trace"), d.get("hbm_read_bytes", 0), d.get("hbm_write_bytes", 0)
    print(f"{os.path.basename(f)[9:-5]:20s} {d['kernel'][:34]:34s} {ns / 1e6:7.3f} {rd / 1e9:6.2f} + {wr / 1e9:5.2f} {(rd + wr) / ns / 1e3:5.2f} "
          f"{d.get('l2_hit_rate', 0):.2f} {d.get('wave_wait_frac', 0):.2f} {d.get('bench_kernel_ms')} {d.get('differs_from_the_unprofiled_line')} {d['commit']}")
