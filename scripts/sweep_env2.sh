#!/bin/bash
# Development aid (GPU box, repo root): bench.py's headline loop under a few settings of one environment knob, interleaved and repeated.
#   scripts/sweep_env2.sh VAR "v1 v2 v3" [bench args]
VAR=$1; VALS=$2; shift; shift
for rep in 1 2; do
  for v in $VALS; do
    ms=$(env $VAR=$v python bench.py --no-cpu --steps 50 --warmup 5 --new-frames 0 "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)
    echo "$VAR=$v rep=$rep ms_per_step=$ms"
  done
done
