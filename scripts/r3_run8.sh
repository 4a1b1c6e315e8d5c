#!/bin/bash
OUT=gpurun_out/r3_run8
mkdir -p $OUT
free -g | head -2
python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default rc $?"; tail -3 $OUT/bench_default.err
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_stream.py -m gpu -x -q -k "deep_seed or sparse or hit_heavy or cfg3a or stream" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
python bench.py --workload mixed --mode locate --no-e2e --steps 2 --warmup 1 > $OUT/bench_mixed_locate.json 2> $OUT/bench_mixed_locate.err
echo "mixed locate rc $?"; tail -3 $OUT/bench_mixed_locate.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-e2e --general-steps 0 --steps 5 --warmup 2 --no-secondary > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob,os,csv
for f in sorted(glob.glob("gpurun_out/r3_run8/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f),"FAILED",e); continue
    r=d["roofline"]
    print(os.path.basename(f), d["value"], d["ms_per_step"], "search", r.get("call",{}).get("ms"), "dom", r["kernel_ms"], "frac", r["frac"], "digests", d["digests"]["status"], d["config"]["locate_kernels_ms"], d["config"]["hits_per_step_rank0"])
    e=d.get("end_to_end") or {}
    print("   e2e", {k:(v.get("value"), v.get("ms")) for k,v in e.items() if isinstance(v,dict)}, "secondary", d.get("secondary") and d["secondary"]["ms_per_step"], "general", d.get("roofline_general") and (d["roofline_general"]["kernel_ms"], d["roofline_general"]["frac"]), "cpu", d.get("cpu_baseline") and (d["cpu_baseline"]["value"], d["cpu_baseline"].get("builds_timed_Mkmers_per_s")))
for f in glob.glob("gpurun_out/r3_run8/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:70]
        if float(r["AverageNs"])>3000 and int(r["Calls"])>=5 and int(r["Calls"])<=40 and "at::" not in n:
            print(f'{float(r["AverageNs"])/1e6:9.3f} ms x{r["Calls"]:>4}  {n}')
PY
