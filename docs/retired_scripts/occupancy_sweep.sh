#!/bin/bash
# orderedSearchKernel's time against the workgroups (of 4 waves) resident per CU: where does more parallelism stop paying?
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
source scripts/ab_lib.sh
for b in 1 2 3 4 5 6; do run occ_$b AWFM_GPU_BLOCKS_PER_CU=$b -- --mode count --no-e2e --no-secondary; done
for b in 6 7 8; do run occ_global_$b AWFM_GPU_PAIR_SUPER=global AWFM_GPU_BLOCKS_PER_CU=$b -- --mode count --no-e2e --no-secondary; done
