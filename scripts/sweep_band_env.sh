#!/bin/bash
# Development aid (GPU box, repo root): the slowest-band time of scripts/band_times.py (world 8, ranks 0 and 3) under settings of one environment knob.
#   scripts/sweep_band_env.sh VAR "v1 v2 v3" [band_times args]
VAR=$1; VALS=$2; shift; shift
for rep in 1 2; do
  for v in $VALS; do
    us=$(env $VAR=$v python scripts/band_times.py --steps 200 --worlds 8 --ranks 0,3 "$@" 2>/dev/null | grep -o '"slowest_us": [0-9.]*' | cut -d' ' -f2)
    echo "$VAR=$v rep=$rep slowest_band_us=$us"
  done
done
