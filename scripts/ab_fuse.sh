#!/bin/bash
# Development aid: plane sum fused into the patch launch (default) vs separate kernel (RPSF_NO_FUSE=1).
for nf in 0 1; do
  if [ $nf = 1 ]; then export RPSF_NO_FUSE=1; else unset RPSF_NO_FUSE; fi
  for cfg in "256 4096" "256 8192" "128 4096" "128 2048" "256 2048"; do
    set -- $cfg
    python3 scripts/kbench.py --n $1 --size $2 --iters 40 --overlap planes --tag "nofuse=$nf"
  done
done
