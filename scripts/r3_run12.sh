#!/bin/bash
OUT=gpurun_out/r3_run12
mkdir -p $OUT
python -m pytest tests/test_gpu_pair.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
Q="--no-cpu --no-e2e --general-steps 0"
python bench.py $Q --steps 20 --warmup 5 > $OUT/locate.json 2> $OUT/locate.err; echo "rc $?"; tail -3 $OUT/locate.err
python bench.py $Q --no-secondary --mode count > $OUT/count.json 2> $OUT/count.err
python bench.py $Q --no-secondary --workload planted --steps 3 > $OUT/planted.json 2> $OUT/planted.err
python bench.py $Q --no-secondary --workload mixed --steps 3 > $OUT/mixed.json 2> $OUT/mixed.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $Q --no-secondary > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob,os,csv
for f in sorted(glob.glob("gpurun_out/r3_run12/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f),"FAILED",e); continue
    r=d["roofline"]
    print(os.path.basename(f), d["value"], d["ms_per_step"], "search", r.get("call",{}).get("ms"), "dom", r["kernel_ms"], "frac", r["frac"], "digests", d["digests"]["status"], d["config"]["locate_kernels_ms"], d.get("secondary") and (d["secondary"]["ms_per_step"], d["secondary"]["with_device_dense_sa"]))
for f in glob.glob("gpurun_out/r3_run12/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:70]
        if float(r["AverageNs"])>3000 and int(r["Calls"])>=5 and int(r["Calls"])<=60 and "at::" not in n:
            print(f'{float(r["AverageNs"])/1e6:9.3f} ms x{r["Calls"]:>4}  {n}')
PY
