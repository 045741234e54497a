#!/bin/bash
# Development aid: start-up stagger sweep (microseconds) on the device-resident kernel benchmark.
for cfg in "256 4096" "256 8192" "128 4096" "128 2048"; do
  set -- $cfg
  for st in 0 8 16 24 32 40 60; do
    python3 scripts/kbench.py --n $1 --size $2 --iters 30 --overlap planes --stagger $st --tag stagger
  done
done
