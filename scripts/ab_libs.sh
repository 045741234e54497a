#!/bin/bash
# Development aid: several builds of the library side by side on the kbench sizes, interleaved (clock / box drift hits all).
#   scripts/ab_libs.sh "<n size> <n size> ..." lib1.so lib2.so ...
CFGS=$1; shift
for rep in 1 2 3; do
  for cfg in $CFGS; do
    n=${cfg%x*}; size=${cfg#*x}
    for lib in "$@"; do
      RPSF_LIB=$lib python3 scripts/kbench.py --n $n --size $size --iters 40 --tag "$(basename $lib .so)"
    done
  done
done
