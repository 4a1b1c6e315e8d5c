#!/bin/bash
# The GPU-box half of refreshing profiles/<round>: every bench line of DESIGN.md 5 and the rocprofv3 runs behind the
# roofline records.  usage (from the repo root, through gpurun): bash scripts/refresh_profiles.sh [tag]
# Afterwards, here: scripts/collect_profiles.py gpurun_out/prof_<name> profiles/<round> <name> "<workload>" for
# name in default ordered_only planted general_pair mixed amino amino_2e9, and copy gpurun_out/bench_<tag>/bench_*.json beside them.
TAG=${1:-r4}
bash scripts/bench_all.sh "$TAG" 2>&1 | tail -20
PROFILE_PASSES="fetch write l2 sq sq2 rdreq" bash scripts/profile_bench.sh default 2>&1 | grep -E "^pass|rc" | tail -8
AWFM_GPU_LOOKUP_FIRST=0 PROFILE_PASSES="fetch write l2 sq rdreq" bash scripts/profile_bench.sh ordered_only 2>&1 | grep -E "^pass" | tail -5
PROFILE_PASSES="fetch write l2" bash scripts/profile_bench.sh planted --workload planted 2>&1 | grep -E "^pass" | tail -4
AWFM_GPU_ORDERED=0 AWFM_GPU_DEEP_SEED_K=0 PROFILE_PASSES="fetch write l2" bash scripts/profile_bench.sh general_pair --mode count 2>&1 | grep -E "^pass" | tail -4
PROFILE_PASSES="fetch write l2" bash scripts/profile_bench.sh mixed --workload mixed 2>&1 | grep -E "^pass" | tail -4
PROFILE_PASSES="fetch write l2" bash scripts/profile_bench.sh amino --alphabet amino 2>&1 | grep -E "^pass" | tail -4
PROFILE_PASSES="fetch write l2" bash scripts/profile_bench.sh amino_2e9 --alphabet amino --text-len 2e9 2>&1 | grep -E "^pass" | tail -4
