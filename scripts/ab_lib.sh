#!/bin/bash
# Development aid: A/B of two builds of the library on the kbench sizes, interleaved so that clock / box drift hits both.
#   scripts/ab_lib.sh <baseline.so> [<candidate.so>]   (candidate defaults to the product library)
BASE=$1; CAND=${2:-regularizepsf_amd/librpsf_hip.so}
for rep in 1 2; do
  for cfg in "256 4096" "256 8192" "128 2048" "256 2048" "128 4096"; do
    set -- $cfg
    RPSF_LIB=$BASE python3 scripts/kbench.py --n $1 --size $2 --iters 40 --tag "base"
    RPSF_LIB=$CAND python3 scripts/kbench.py --n $1 --size $2 --iters 40 --tag "cand"
  done
done
