#!/bin/bash
# Development aid: rim-first processing order on / off.
for nr in 0 1; do
  if [ $nr = 1 ]; then export RPSF_NO_RIM_FIRST=1; else unset RPSF_NO_RIM_FIRST; fi
  for cfg in "256 4096" "256 8192" "128 4096" "128 2048" "256 2048"; do
    set -- $cfg
    python3 scripts/kbench.py --n $1 --size $2 --iters 40 --overlap planes --tag "norimfirst=$nr"
  done
done
