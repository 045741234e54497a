#!/bin/bash
# Development aid: A/B of two builds of the library with bench.py's own timed loop (no CPU leg, no verification), interleaved so that clock and box
# drift hit both.   scripts/ab_bench.sh <baseline.so> [reps]      (candidate = the product library)
BASE=$(realpath $1); REPS=${2:-3}
one() {  # $1 lib (empty: product), $2 label, $3... bench args
  local lib=$1 label=$2; shift 2
  RPSF_LIB=$lib python3 bench.py "$@" --steps 50 --no-cpu --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step'], d['roofline']['frac'])"
}
for rep in $(seq $REPS); do
  for cfg in "3 --no-e2e --new-frames 0" "2 --no-e2e --new-frames 0" "5" "4 --no-e2e --new-frames 0"; do
    set -- $cfg; c=$1; shift
    one "$BASE" "config$c base" --config $c "$@"
    one "" "config$c cand" --config $c "$@"
  done
done
