#!/usr/bin/env python3
"""one-screen summary of a bench.py JSON line (usage: bench_summary.py <file>)"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print("value", d["value"], "ms/step", d["ms_per_step"], "| front:", c.get("lookup_front"), "| tail:", c.get("list_tail"))
r = d["roofline"]
print("roofline", r["kernel"], "frac", r["frac"], "kernel_ms", r["kernel_ms"], "hbm_measured", r.get("hbm_frac_measured"), "ref_alg", (r.get("reference_algorithm") or {}).get("frac"), (r.get("reference_algorithm") or {}).get("kernel_ms"))
for k in ("search_call_ms", "locate_kernels_ms", "index_build_s", "device_seed_build_s", "planted_ms_per_step", "planted_lf_walk_ms_per_step", "mixed_lengths_ms_per_step",
          "amino_value", "amino_ms_per_step", "amino_roofline_frac", "amino_2e9_value", "amino_2e9_ms_per_step", "amino_2e9_roofline_frac", "repetitive_value", "repetitive_ms_per_step", "repetitive_roofline_frac", "dense_form_ms_per_step", "device_image_bytes"):
    if k in c:
        print(" ", k, c[k])
sp = d.get("scaling_proxy")
if sp:
    print("proxy whole", sp["whole_batch_ms"], "planted whole", sp.get("planted_whole_batch_ms"))
    for name, v in sp["shards"].items():
        print("  ", name, {n: (x["ms_max"], x["efficiency"]) for n, x in v.items()})
if d.get("cpu_baseline"):
    print("cpu", d["cpu_baseline"]["value"], "cores", d["cpu_baseline"]["cores"])
e = d.get("end_to_end") or {}
for k, v in e.items():
    if isinstance(v, dict) and "Mkmers_per_s" in v:
        print("  e2e", k, v["Mkmers_per_s"], v.get("ms"))
