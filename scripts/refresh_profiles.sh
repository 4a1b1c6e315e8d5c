#!/bin/bash
# The GPU-box half of refreshing profiles/<round>: every bench line of DESIGN.md 5 and the rocprofv3 runs behind the
# roofline records, ALL from one tree.  usage (from the repo root, through gpurun):
#   AWFM_COMMIT=$(git rev-parse --short HEAD) -> gpurun -- 'AWFM_COMMIT=<hash> bash scripts/refresh_profiles.sh r5'
# Afterwards, here: scripts/collect_all_profiles.sh r5 (collect_profiles.py per set, which refuses a set whose kernel-trace
# average is not the bench line's kernel_ms), and copy gpurun_out/bench_<tag>/bench_*.json beside them.
TAG=${1:-r6}
export AWFM_COMMIT=${AWFM_COMMIT:-unknown}
if [ "${SKIP_BENCH_ALL:-0}" != "1" ]; then bash scripts/bench_all.sh "$TAG" 2>&1 | tail -45; fi
# SETS="planted mixed" limits the rocprofv3 part to those sets (a call that ran out of time is continued, SKIP_BENCH_ALL=1).
P="fetch write l2 sq"
SETS=${SETS:-default wide wide_planted ordered_only planted planted_count general_pair exact_tables mixed amino amino_2e9 amino_wide repetitive_unique repetitive_planted shards}
want() { case " $SETS " in *" $1 "*) return 0;; esac; return 1; }
prof() { bash scripts/profile_bench.sh "$@" 2>&1 | grep -E "^pass|rc" | tail -8; }
want default && PROFILE_PASSES="fetch write l2 sq sq2 rdreq" prof default
want wide && PROFILE_PASSES="$P rdreq" prof wide --text-len 6.2e9 --no-wide
want wide_planted && PROFILE_PASSES="$P" prof wide_planted --text-len 6.2e9 --workload planted --no-wide
want ordered_only && AWFM_GPU_LOOKUP_FIRST=0 PROFILE_PASSES="$P rdreq" prof ordered_only
want planted && PROFILE_PASSES="$P" prof planted --workload planted
want planted_count && PROFILE_PASSES="$P" prof planted_count --workload planted --mode count
want general_pair && AWFM_GPU_ORDERED=0 AWFM_GPU_DEEP_SEED_K=0 PROFILE_PASSES="$P" prof general_pair --mode count
want exact_tables && AWFM_GPU_ORDERED=0 PROFILE_PASSES="$P" prof exact_tables --mode count
want mixed && PROFILE_PASSES="$P" prof mixed --workload mixed
want amino && PROFILE_PASSES="$P" prof amino --alphabet amino
want amino_2e9 && PROFILE_PASSES="$P" prof amino_2e9 --alphabet amino --text-len 2e9
want amino_wide && PROFILE_PASSES="$P" prof amino_wide --alphabet amino --text-len 4.4e9
want repetitive_unique && PROFILE_PASSES="$P" prof repetitive_unique --text repetitive --workload unique
want repetitive_planted && PROFILE_PASSES="$P" prof repetitive_planted --text repetitive --workload planted
# the shard-sized step (what a rank of an 8-GPU strong run does): kernel-trace timelines
want shards && bash scripts/r5_trace_shards.sh shards_"$TAG" > /dev/null 2>&1
exit 0
