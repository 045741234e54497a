#!/bin/bash
# Development aid: A/B of kernel generation (RPSF_V1=1 selects the first-generation three-stage kernels) and overlap mode.
for gen in v2 v1; do
  for mode in planes direct; do
    for cfg in "256 4096" "128 2048" "256 8192" "128 4096"; do
      set -- $cfg
      if [ $gen = v1 ]; then export RPSF_V1=1; else unset RPSF_V1; fi
      python3 scripts/kbench.py --n $1 --size $2 --iters 40 --overlap $mode --tag $gen-$mode
    done
  done
done
