#!/bin/bash
# where the seed-order path starts to pay against the general kernel (pair steps): search time by batch size
cd "$(dirname "$0")/.."
source scripts/ab_lib.sh
for q in 4e6 8e6 16e6 32e6 64e6; do
  for w in random planted; do
    run thr_${w}_${q}_general AWFM_GPU_ORDERED=0 -- --mode count --no-e2e --queries $q --workload $w
    run thr_${w}_${q}_ordered AWFM_GPU_ORDERED=1 -- --mode count --no-e2e --queries $q --workload $w
  done
done
