#!/bin/bash
# Here, after scripts/refresh_profiles.sh ran on a GPU box: condense every profiled set into profiles/<round>/ (a set whose
# kernel-trace average is not its bench line's kernel_ms is refused and named), copy the bench lines and the shard timelines.
R=${1:-r6}
D=profiles/$R
mkdir -p "$D"
B=gpurun_out/bench_$R
collect() { # set, workload text, un-profiled bench line
  python3 scripts/collect_profiles.py "gpurun_out/prof_$1" "$D" "$1" "$2" "$B/bench_$3.json" || echo "== set $1 NOT collected"
}
collect default "100 M random 21-mers, locate, 3.1 Gbp" default
collect wide "100 M random 21-mers, locate, 6.2 Gbp (an index beyond 2^32 positions: 64-bit instantiations)" wide
collect wide_planted "100 M planted 21-mers, locate, 6.2 Gbp" wide_planted
collect ordered_only "100 M random 21-mers, locate, 3.1 Gbp, AWFM_GPU_LOOKUP_FIRST=0" no_lookup_first
collect planted "100 M planted 21-mers, locate, 3.1 Gbp" planted
collect planted_count "100 M planted 21-mers, count, 3.1 Gbp (counts home from search order: countScatterKernel, countPlaceKernel)" planted_count
collect general_pair "100 M random 21-mers, count, 3.1 Gbp, AWFM_GPU_ORDERED=0 AWFM_GPU_DEEP_SEED_K=0 (exact-range general kernel)" general
collect exact_tables "100 M random 21-mers, count, 3.1 Gbp, AWFM_GPU_ORDERED=0 (awfmGpuSearch through the tables: exactLookupSearchKernel)" exact_tables
collect mixed "100 M mixed 8..30-mers, count, 3.1 Gbp" mixed
collect amino "50 M random 10-mers, locate, 200 M residues" amino
collect amino_2e9 "50 M random 10-mers, locate, 2 G residues" amino_2e9
collect amino_wide "50 M random 10-mers, locate, 4.4 G residues (64-bit instantiations)" amino_wide
collect repetitive_unique "100 M 21-mers from the unique sequence of a genome-shaped 3.1 Gbp text, locate" repetitive_unique
collect repetitive_planted "100 M planted 21-mers, count, genome-shaped 3.1 Gbp text" repetitive_planted
cp "$B"/bench_*.json "$D"/ 2>/dev/null
mkdir -p "$D/shard_timelines"
cp gpurun_out/shards_"$R"/*_gaps.txt gpurun_out/shards_"$R"/*.log "$D/shard_timelines/" 2>/dev/null
ls "$D" | wc -l
