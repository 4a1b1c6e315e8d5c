#!/bin/bash
# A/B the search kernel variants on the default bench workload (count mode, no CPU leg)
for k in group8 pipe1 pipe2 pipe4; do
  for b in "" 1 2 3 4 5 6 7 8; do
    [ -n "$b" ] && export AWFM_GPU_BLOCKS_PER_CU=$b || unset AWFM_GPU_BLOCKS_PER_CU
    AWFM_GPU_KERNEL=$k timeout 200 python bench.py --no-cpu --mode count --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$k blocks/CU=${b:-auto}', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
