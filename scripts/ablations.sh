#!/bin/bash
# Development aid: time the ablation builds of the second-generation kernel (devlibs/abl_*.so, results are wrong by design).
for lib in "" NOGATHER NOK NOSTORE NOLDS NOLDS_NOBAR NOVALU NOGATHER_NOK_NOSTORE; do
  if [ -n "$lib" ]; then export RPSF_LIB=devlibs/abl_$lib.so; else unset RPSF_LIB; fi
  for cfg in "256 8192" "256 4096" "128 4096"; do
    set -- $cfg
    python3 scripts/kbench.py --n $1 --size $2 --iters 20 --overlap planes --tag "abl:${lib:-none}"
  done
done
