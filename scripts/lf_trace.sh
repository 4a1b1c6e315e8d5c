#!/bin/bash
OUT=$PWD/gpurun_out/lf_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
one() { # tag, env..., -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
rm -rf $OUT/trace_$tag
env "${envs[@]}" rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-e2e --no-secondary --general-steps 0 --steps 4 --warmup 2 "$@" > $OUT/bench_$tag.log 2>&1
python3 - <<PY
import csv,glob,re
f=glob.glob("$OUT/trace_$tag/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "encodeLookupKernel" in r["Kernel_Name"]]
i,j=idx[-2],idx[-1]
out=[]
for r in rows[i:j]:
    n=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); n=re.sub(r"\(.*","",n)[:40]
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    if d>0.02: out.append(f"{n.split('<')[0].replace('void ','')} {d:.3f}")
print("$tag:", "; ".join(out), "| step span", (int(rows[j]["Start_Timestamp"])-int(rows[i]["Start_Timestamp"]))/1e6)
PY
}
one count -- --mode count
one list --
one list_b4 AWFM_GPU_BLOCKS_PER_CU=4 --
one list_c1 AWFM_GPU_CHUNKS_PER_TICKET=1 --
one list_c16 AWFM_GPU_CHUNKS_PER_TICKET=16 --
