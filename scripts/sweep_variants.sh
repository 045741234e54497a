#!/bin/bash
# Development aid: scripts/sweep_times.py for several builds of the library (devlibs/librpsf_<name>.so), interleaved and repeated.
#   scripts/sweep_variants.sh "<cases>" name1 name2 ...   ("product" = the product library)
CASES=$1; shift
for name in "$@"; do
  lib=$PWD/devlibs/librpsf_$name.so; [ "$name" = product ] && lib=$PWD/regularizepsf_amd/librpsf_hip.so
  RPSF_LIB=$lib timeout 120 python3 scripts/sweep_check.py 2>&1 | tr '\n' ';'; echo " <- $name"
done
for rep in 1 2 3; do
  for name in "$@"; do
    lib=$PWD/devlibs/librpsf_$name.so; [ "$name" = product ] && lib=$PWD/regularizepsf_amd/librpsf_hip.so
    RPSF_LIB=$lib timeout 120 python3 scripts/sweep_times.py --cases $CASES --iters 40 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print('$name', 'rep$rep', d['n'], d['size'], 'med', d['ms_med'], 'min', d['ms_min'], 'frac', d['frac'])"
  done
done
