#!/bin/bash
# Development aid: number of summing workgroups that run beside the patches from the start of the fused launch.
for sf in 0 16 32; do
  export RPSF_SUM_FIRST=$sf
  for cfg in "256 4096" "256 8192" "256 2048"; do
    set -- $cfg
    python3 scripts/kbench.py --n $1 --size $2 --iters 40 --overlap planes --tag "sum_first=$sf"
  done
done
