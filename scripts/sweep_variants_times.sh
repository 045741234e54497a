#!/bin/bash
# Development aid: scripts/sweep_times.py only (no correctness check: ablation builds give wrong results on purpose), interleaved and repeated.
CASES=$1; shift
for rep in 1 2; do
  for name in "$@"; do
    lib=$PWD/devlibs/librpsf_$name.so; [ "$name" = product ] && lib=$PWD/regularizepsf_amd/librpsf_hip.so
    RPSF_LIB=$lib timeout 120 python3 scripts/sweep_times.py --cases $CASES --iters 40 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('$name', 'rep$rep', d['n'], d['size'], 'med', d['ms_med'], 'min', d['ms_min'], 'frac', d['frac'])"
  done
done
