#!/bin/bash
OUT=gpurun_out/r3_run7
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default rc $?"; tail -3 $OUT/bench_default.err
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
python bench.py --workload planted --no-cpu --steps 3 --warmup 1 > $OUT/bench_planted.json 2> $OUT/bench_planted.err
python bench.py --workload mixed --no-cpu --steps 3 --warmup 1 > $OUT/bench_mixed.json 2> $OUT/bench_mixed.err
python bench.py --mode count --no-cpu --steps 5 --warmup 2 > $OUT/bench_count.json 2> $OUT/bench_count.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r3_run7/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f),"FAILED",e); continue
    r=d["roofline"]
    print(os.path.basename(f), d["value"], d["ms_per_step"], "search", r.get("call",{}).get("ms"), "dom", r["kernel_ms"], "frac", r["frac"], "digests", d["digests"]["status"], d["config"]["locate_kernels_ms"])
    e=d.get("end_to_end") or {}
    print("   e2e", {k:(v.get("value"), v.get("ms")) for k,v in e.items() if isinstance(v,dict)}, "secondary", d.get("secondary") and d["secondary"]["ms_per_step"], "general", d.get("roofline_general") and (d["roofline_general"]["kernel_ms"], d["roofline_general"]["frac"]))
PY
