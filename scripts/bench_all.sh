#!/bin/bash
# every bench.py configuration quoted in DESIGN.md, one JSON line each -> gpurun_out/bench_<tag>/bench_<name>.json
TAG=${1:-r6}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/bench_$TAG
mkdir -p "$OUT"
cd "$ROOT"
# $ONLY="name name ...": those lines only (a kernel changed: its lines again, the others stand)
run() { # name [ENV=..]... -- bench args
  name=$1; shift
  if [ -n "${ONLY:-}" ]; then case " $ONLY " in *" $name "*) ;; *) return 0;; esac; fi
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 bench.py "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/bench_$name.json"
}
run default -- --steps 20 --warmup 5
run count -- --mode count
run planted -- --workload planted
run planted_dense_results AWFM_BENCH_DENSE_RESULTS=1 -- --workload planted --no-cpu --no-e2e
run mixed -- --workload mixed
run mixed_no_lookup AWFM_GPU_MIXED_LOOKUP=0 -- --workload mixed --no-cpu --no-e2e --general-steps 0
run mixed_no_prediction AWFM_GPU_LOOKUP_PREDICT=0 -- --workload mixed --no-cpu --no-e2e --general-steps 0
run mixed_short -- --workload mixed --mixed-lengths 8 15 --no-cpu --no-e2e --general-steps 0 --no-shard-proxy
run mixed_long -- --workload mixed --mixed-lengths 18 30 --no-cpu --no-e2e --general-steps 0 --no-shard-proxy
run mixed_locate -- --workload mixed --mode locate --no-e2e --steps 2 --warmup 1
run mixed_locate_16_30 -- --workload mixed --mixed-lengths 16 30 --mode locate --no-e2e --no-secondary --no-dense-form
run amino -- --alphabet amino
run amino_2e9 -- --alphabet amino --text-len 2e9 --no-e2e
run amino_no_lookup AWFM_GPU_AMINO_LOOKUP=0 -- --alphabet amino --no-cpu --no-e2e
run amino_no_deep_table AWFM_GPU_AMINO_DEEP_SEED_K=0 -- --alphabet amino --no-cpu --no-e2e
run weak -- --scaling weak --no-cpu --no-e2e --no-secondary --general-steps 0 --no-shard-proxy
run general AWFM_GPU_ORDERED=0 AWFM_GPU_DEEP_SEED_K=0 -- --mode count --no-cpu --no-e2e
run dense_results AWFM_BENCH_DENSE_RESULTS=1 -- --no-cpu --no-e2e --no-secondary --general-steps 0
run no_deep_table -- --device-seed-k 0 --no-cpu --no-e2e --no-secondary --general-steps 0
run deep_table_14 -- --device-seed-k 14 --no-cpu --no-e2e --no-secondary --general-steps 0
run no_next_bits AWFM_GPU_DEEP_NEXT=0 -- --no-cpu --no-e2e --no-secondary --general-steps 0
run no_lookup_first AWFM_GPU_LOOKUP_FIRST=0 -- --no-cpu --no-e2e --no-secondary --general-steps 0
run no_lookup_first_count AWFM_GPU_LOOKUP_FIRST=0 -- --mode count --no-cpu --no-e2e --no-secondary --general-steps 0
run nopair_default AWFM_GPU_PAIR=0 -- --no-cpu --no-e2e --no-secondary --general-steps 0
run planted_lf_walk -- --workload planted --no-device-dense-sa --no-cpu --no-e2e
run lf_walk -- --no-device-dense-sa --no-cpu --no-e2e --no-secondary --general-steps 0
run repetitive_random -- --text repetitive --no-cpu --no-e2e --general-steps 0
run repetitive_planted -- --text repetitive --workload planted --no-cpu --no-e2e --general-steps 0
run repetitive_unique -- --text repetitive --workload unique --no-cpu --no-e2e --general-steps 0
# round 5
run no_prediction AWFM_GPU_LOOKUP_PREDICT=0 -- --no-cpu --no-e2e --no-secondary --general-steps 0
run exact_tables AWFM_GPU_ORDERED=0 -- --mode count --no-cpu --no-e2e --no-secondary
run exact_tables_off AWFM_GPU_ORDERED=0 AWFM_GPU_EXACT_LOOKUP=0 -- --mode count --no-cpu --no-e2e --no-secondary
run amino_no_next_bits AWFM_GPU_DEEP_NEXT=0 -- --alphabet amino --no-cpu --no-e2e
run amino_2e9_no_next_bits AWFM_GPU_DEEP_NEXT=0 -- --alphabet amino --text-len 2e9 --no-cpu --no-e2e
run amino_planted -- --alphabet amino --workload planted --no-cpu --no-e2e
# round 6: an index beyond 2^32 positions as the main line (the 64-bit instantiations), random and planted
run wide -- --text-len 6.2e9 --no-e2e --no-amino --no-repetitive --no-wide
run wide_planted -- --text-len 6.2e9 --workload planted --no-cpu --no-e2e --no-wide
run wide_mixed -- --text-len 6.2e9 --workload mixed --no-cpu --no-e2e --general-steps 0 --no-wide
run amino_wide -- --alphabet amino --text-len 4.4e9 --no-cpu --no-e2e --no-secondary --no-shard-proxy
# counts only of a dense-hit batch: the counts come home from search order in whole lines (awfm_count_order_kernel.h)
run planted_count -- --workload planted --mode count --no-cpu --no-e2e --no-secondary --general-steps 0 --no-shard-proxy
python3 - "$OUT" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
        continue
    r, c = d["roofline"], d.get("cpu_baseline") or {}
    print(f"{os.path.basename(f)[6:-5]:22s} {d['value']:9.1f} Mkmers/s  {d['ms_per_step']:7.2f} ms/step  search {r['kernel_ms']:6.2f} ms  "
          f"frac {r['frac']:.3f}  locate {d['config'].get('locate_kernels_ms', 0):6.2f} ms  cpu {c.get('value')} on {c.get('cores')}")
PY
