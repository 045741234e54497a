#!/bin/bash
# Development aid: A/B of the overlap-add strategies on the device-resident kernel benchmark.
for mode in direct planes; do
  for cfg in "256 4096" "128 2048" "256 8192" "128 4096"; do
    set -- $cfg
    python3 scripts/kbench.py --n $1 --size $2 --iters 40 --overlap $mode --tag $mode
  done
done
