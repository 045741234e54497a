#!/bin/bash
# Development aid: tail split (partial last round + plane sum in one launch) on / off.
for ns in 0 1; do
  if [ $ns = 1 ]; then export RPSF_NO_SPLIT=1; else unset RPSF_NO_SPLIT; fi
  for cfg in "256 4096" "256 8192" "128 4096" "128 2048" "256 2048"; do
    set -- $cfg
    python3 scripts/kbench.py --n $1 --size $2 --iters 40 --overlap planes --tag "nosplit=$ns"
  done
done
