#!/bin/bash
OUT=gpurun_out/r3_run16
mkdir -p $OUT
python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "cfg3b" > $OUT/pytest_cfg3b.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_cfg3b.log; tail -4 $OUT/pytest_cfg3b.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_fuzz.py -m gpu -x -q -k "ordered or bucketed or packed or stream or fuzz or sparse or search_order" > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
Q="--no-cpu --no-e2e --general-steps 0 --no-secondary"
for i in 1 2; do
python bench.py $Q --steps 10 --warmup 3 > $OUT/locate_$i.json 2> $OUT/locate_$i.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/locate_$i.json") if l.startswith("{")][-1]); r=d["roofline"]
print("locate", d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], d["digests"]["status"])
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $Q > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "partitionKernel|encodeCodes" --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py $Q --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob,csv,collections
for f in glob.glob("gpurun_out/r3_run16/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:70]
        if any(k in n for k in ("partitionKernel","encodeCodes4","orderedSearchKernel<4, true, true, false, true, false, true","bucketScan")):
            print(f'{float(r["AverageNs"])/1e6:9.3f} ms x{r["Calls"]:>4}  {n}')
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/r3_run16/pmc_write/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, "WRITE_SIZE KB mean", sum(v)/len(v))
PY
