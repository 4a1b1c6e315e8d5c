#!/bin/bash
OUT=gpurun_out/r3_run9
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for probe in 0 1 2; do
AWFM_GPU_PARTITION_PROBE=$probe rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace$probe -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-e2e --general-steps 0 --steps 5 --warmup 2 --no-secondary --mode count > $GRAFT_REPO_ROOT/$OUT/trace$probe.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob,csv
for p in (0,1,2):
  for f in glob.glob(f"gpurun_out/r3_run9/trace{p}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:60]
        if any(k in n for k in ("partitionKernel","encodeCodes4","orderedSearchKernel<4, true, true, false, true, false, true")):
            print(p, f'{float(r["AverageNs"])/1e6:9.3f} ms x{r["Calls"]:>4}  {n}')
PY
