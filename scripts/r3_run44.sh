#!/bin/bash
OUT=$PWD/gpurun_out/r3_run44
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deep_seed or ordered_hits_only_search_is_exact" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-e2e --no-secondary --general-steps 0 --steps 5 --warmup 2 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('default', d['value'], d['ms_per_step'], 'search', r['call']['ms'], 'dom', r['kernel_ms'], d['digests']['status'], 'build', d['config']['index_build_s'])"
python3 - <<PY
import csv,glob,re
f=glob.glob("$OUT/trace/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=re.sub(r"\(anonymous namespace\)::","",r["Name"]); n=re.sub(r"\(.*","",n)[:70]
    if any(x in n for x in ("encodeCodes","partitionKernel","orderedSearch","deepNext","deepSeedLevel","deepBig")):
        print(f"{n:70s} {r['Calls']:>4s} {float(r['AverageNs'])/1e6:9.3f} ms")
PY
