#!/bin/bash
# Development aid: direct overlap-add vs planes, and processing-order variants (run length of the colour sort).
for run in default 100000 128 64; do
  if [ $run = default ]; then unset RPSF_ORDER_RUN; else export RPSF_ORDER_RUN=$run; fi
  for mode in direct planes; do
    for cfg in "256 4096" "256 8192" "128 4096"; do
      set -- $cfg
      python3 scripts/kbench.py --n $1 --size $2 --iters 30 --overlap $mode --tag "run=$run:$mode"
    done
  done
done
