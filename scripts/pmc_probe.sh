#!/bin/bash
# PMC passes over scripts/sorted_probe.py (unsorted vs seed-sorted batch): 4 dispatches each, in that order
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PROBE_SHORT=1
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-include-regex "searchKernel" --output-format csv -d $OUT/p$i -- python3 $ROOT/scripts/sorted_probe.py > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    d.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in d.items():
    print(k, " ".join(f"{n}={x:.4g}" for n, x in v.items()))
PY
done
