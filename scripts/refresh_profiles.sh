#!/bin/bash
# The GPU-box half of refreshing profiles/<round>: every bench line of DESIGN.md 5 and the rocprofv3 runs behind the
# roofline records, ALL from one tree.  usage (from the repo root, through gpurun):
#   AWFM_COMMIT=$(git rev-parse --short HEAD) -> gpurun -- 'AWFM_COMMIT=<hash> bash scripts/refresh_profiles.sh r5'
# Afterwards, here: scripts/collect_all_profiles.sh r5 (collect_profiles.py per set, which refuses a set whose kernel-trace
# average is not the bench line's kernel_ms), and copy gpurun_out/bench_<tag>/bench_*.json beside them.
TAG=${1:-r5}
export AWFM_COMMIT=${AWFM_COMMIT:-unknown}
if [ "${SKIP_BENCH_ALL:-0}" != "1" ]; then bash scripts/bench_all.sh "$TAG" 2>&1 | tail -45; fi
P="fetch write l2 sq"
PROFILE_PASSES="fetch write l2 sq sq2 rdreq" bash scripts/profile_bench.sh default 2>&1 | grep -E "^pass|rc" | tail -8
AWFM_GPU_LOOKUP_FIRST=0 PROFILE_PASSES="$P rdreq" bash scripts/profile_bench.sh ordered_only 2>&1 | grep -E "^pass" | tail -5
PROFILE_PASSES="$P" bash scripts/profile_bench.sh planted --workload planted 2>&1 | grep -E "^pass" | tail -4
AWFM_GPU_ORDERED=0 AWFM_GPU_DEEP_SEED_K=0 PROFILE_PASSES="$P" bash scripts/profile_bench.sh general_pair --mode count 2>&1 | grep -E "^pass" | tail -4
AWFM_GPU_ORDERED=0 PROFILE_PASSES="$P" bash scripts/profile_bench.sh exact_tables --mode count 2>&1 | grep -E "^pass" | tail -4
PROFILE_PASSES="$P" bash scripts/profile_bench.sh mixed --workload mixed 2>&1 | grep -E "^pass" | tail -4
PROFILE_PASSES="$P" bash scripts/profile_bench.sh amino --alphabet amino 2>&1 | grep -E "^pass" | tail -4
PROFILE_PASSES="$P" bash scripts/profile_bench.sh amino_2e9 --alphabet amino --text-len 2e9 2>&1 | grep -E "^pass" | tail -4
PROFILE_PASSES="$P" bash scripts/profile_bench.sh repetitive_unique --text repetitive --workload unique 2>&1 | grep -E "^pass" | tail -4
PROFILE_PASSES="$P" bash scripts/profile_bench.sh repetitive_planted --text repetitive --workload planted 2>&1 | grep -E "^pass" | tail -4
# the shard-sized step (what a rank of an 8-GPU strong run does): kernel-trace timelines
bash scripts/r5_trace_shards.sh shards_"$TAG" > /dev/null 2>&1
