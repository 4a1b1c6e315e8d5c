#!/bin/bash
# the AoS entry points' chunk size and lanes, one box
out=gpurun_out/r5_aos_ab
mkdir -p $out
for v in "default" "AWFM_GPU_AOS_CHUNK=524288" "AWFM_GPU_AOS_CHUNK=262144" "AWFM_GPU_AOS_CHUNK=2097152" "AWFM_GPU_DEVICES=0,0,0,0" "AWFM_GPU_DEVICES=0,0,0,0 AWFM_GPU_AOS_CHUNK=524288"; do
  name=$(echo "$v" | tr ' =,' '___')
  if [ "$v" = "default" ]; then envs=(); else envs=($v); fi
  env "${envs[@]}" python bench.py --no-cpu --no-secondary --no-shard-proxy --general-steps 0 --steps 3 --warmup 1 --no-dense-form > $out/$name.json 2> $out/$name.err
  python - "$out/$name.json" "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    e=d["end_to_end"]
    a,p=e["aos_drop_in"],e["aos_drop_in_planted"]
    print(sys.argv[2], "| random", a["value"], a["ms_all"], a["host"]["frac_of_host_bound"], a["host"]["stage_ms_summed_over_chunks"], "| planted", p["value"], p["ms_all"], p["host"]["frac_of_host_bound"])
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
done
