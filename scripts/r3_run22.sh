#!/bin/bash
OUT=gpurun_out/r3_run22
mkdir -p $OUT
show() {
python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
    e=d.get("end_to_end") or {}
    print(sys.argv[2], d["value"], d["ms_per_step"], "search", r["call"]["ms"], "dom", r["kernel_ms"], "frac", r["frac"], d["digests"]["status"], "locate", d["config"].get("locate_kernels_ms"), "build", d["config"]["device_seed_build_s"], {k:(v.get("Mkmers_per_s") if isinstance(v,dict) else v) for k,v in e.items()} if e else "", (d.get("secondary") or {}).get("ms_per_step"))
except Exception as ex:
    print(sys.argv[2], "failed", ex, open(sys.argv[1][:-5]+".err").read()[-600:])
PY
}
for k in 16 14; do
python bench.py --device-seed-k $k --no-cpu --no-secondary --general-steps 0 --steps 10 --warmup 3 > $OUT/default_$k.json 2> $OUT/default_$k.err; show $OUT/default_$k.json "default $k"
python bench.py --device-seed-k $k --workload planted --no-cpu --no-e2e --general-steps 0 --steps 5 --warmup 2 > $OUT/planted_$k.json 2> $OUT/planted_$k.err; show $OUT/planted_$k.json "planted $k"
python bench.py --device-seed-k $k --workload mixed --no-cpu --no-e2e --general-steps 0 --no-secondary --steps 5 --warmup 2 > $OUT/mixed_$k.json 2> $OUT/mixed_$k.err; show $OUT/mixed_$k.json "mixed $k"
python bench.py --device-seed-k $k --workload planted --device-dense-sa --no-cpu --no-e2e --general-steps 0 --steps 5 --warmup 2 > $OUT/planted_dsa_$k.json 2> $OUT/planted_dsa_$k.err; show $OUT/planted_dsa_$k.json "planted dense sa $k"
done
rocm-smi --showmeminfo vram | head -8
