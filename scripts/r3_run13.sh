#!/bin/bash
OUT=gpurun_out/r3_run13
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "search_order or sparse_hit_list" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
Q="--no-cpu --no-e2e --general-steps 0"
python bench.py $Q --no-secondary --workload planted --steps 3 > $OUT/planted_order.json 2> $OUT/planted_order.err; echo "rc $?"; tail -3 $OUT/planted_order.err
AWFM_BENCH_DENSE_RESULTS=1 python bench.py $Q --no-secondary --workload planted --steps 3 > $OUT/planted_dense.json 2> $OUT/planted_dense.err
python bench.py $Q --steps 5 --warmup 2 > $OUT/locate.json 2> $OUT/locate.err; echo "rc $?"; tail -3 $OUT/locate.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $Q --no-secondary --workload planted --steps 3 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob,os,csv
for f in sorted(glob.glob("gpurun_out/r3_run13/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f),"FAILED",e); continue
    r=d["roofline"]
    print(os.path.basename(f), d["value"], d["ms_per_step"], "search", r.get("call",{}).get("ms"), "dom", r["kernel_ms"], "frac", r["frac"], "digests", d["digests"]["status"], d["config"]["locate_kernels_ms"], d["config"]["search_path"], d.get("secondary") and (d["secondary"]["ms_per_step"], d["secondary"]["with_device_dense_sa"]))
for f in glob.glob("gpurun_out/r3_run13/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:70]
        if float(r["AverageNs"])>30000 and int(r["Calls"])>=3 and int(r["Calls"])<=60 and "at::" not in n:
            print(f'{float(r["AverageNs"])/1e6:9.3f} ms x{r["Calls"]:>4}  {n}')
PY
