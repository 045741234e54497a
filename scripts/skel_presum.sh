#!/bin/bash
# Development aid: config 5 / config 2 step times of the product and of the pre-sum timing skeleton (devlibs/librpsf_skelpresum.so: wrong results on purpose)
for rep in 1 2 3; do
  for lib in regularizepsf_amd/librpsf_hip.so devlibs/librpsf_skelpresum.so; do
    for cfg in 5 2; do
      line=$(RPSF_LIB=$PWD/$lib timeout 300 python3 bench.py --config $cfg --no-cpu --no-verify --no-e2e 2>/dev/null | tail -1)
      echo "$(basename $lib .so) rep=$rep config=$cfg $(echo "$line" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'])")"
    done
  done
done
