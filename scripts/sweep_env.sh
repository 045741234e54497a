#!/bin/bash
# Development aid: bench.py's ms_per_step for several values of one environment variable, interleaved and repeated.
#   scripts/sweep_env.sh RPSF_SUM_FIRST "0 8 16 32" [bench args...]
VAR=$1; VALS=$2; shift; shift
for rep in 1 2 3; do
  for v in $VALS; do
    ms=$(env $VAR=$v python3 bench.py --no-cpu "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)
    echo "$VAR=$v rep=$rep ms_per_step=$ms"
  done
done
