#!/bin/bash
OUT=$PWD/gpurun_out/r3_run43
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-e2e --no-secondary --general-steps 0 --steps 5 --warmup 2 > $OUT/bench.log 2>&1
python3 - <<PY
import csv,glob,re
f=glob.glob("$OUT/trace/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=re.sub(r"\(anonymous namespace\)::","",r["Name"]); n=re.sub(r"\(.*","",n)[:70]
    if any(x in n for x in ("encodeCodes","partitionKernel","orderedSearch","bucketScan","fillSparse","deepNext","deepSeedLevel","deepBig")):
        print(f"{n:70s} {r['Calls']:>4s} {float(r['AverageNs'])/1e6:9.3f} ms")
PY
