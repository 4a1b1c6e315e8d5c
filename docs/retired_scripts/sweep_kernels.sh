#!/bin/bash
# A/B the search kernel variants on the default bench workload (count mode, no CPU leg), one box, interleaved
KERNELS=${KERNELS:-"g4 g2"}
BLOCKS=${BLOCKS:-"auto 4 5 6 7 8"}
for rep in 1 2; do
for k in $KERNELS; do
  for b in $BLOCKS; do
    [ "$b" != auto ] && export AWFM_GPU_BLOCKS_PER_CU=$b || unset AWFM_GPU_BLOCKS_PER_CU
    AWFM_GPU_KERNEL=$k timeout 200 python bench.py --no-cpu --mode count --steps 5 --warmup 2 "$@" 2>&1 | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep$rep $k blocks/CU=$b', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
done
