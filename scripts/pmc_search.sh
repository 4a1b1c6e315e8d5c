#!/bin/bash
# PMC pass over the search kernel only (count mode): scripts/pmc_search.sh <tag> <kernel> "<counters>"
TAG=$1; KER=$2; CNT=$3
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
export AWFM_GPU_KERNEL=$KER
rocprofv3 --pmc $CNT --kernel-include-regex "search" --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --no-cpu --mode count --steps 2 --warmup 1 > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
for f in glob.glob(sys.argv[1]+'/*/*counter_collection.csv'):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'Tally' in r['Kernel_Name'] or 'Lb1EEEv' in r['Kernel_Name'][-60:]: pass
        agg[(r['Kernel_Name'].split('(')[0][-45:],r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in sorted(agg.items()): print(k,c,len(v),'%.4g'%(sum(v)/len(v)))
PY
