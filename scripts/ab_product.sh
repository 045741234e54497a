#!/bin/bash
# Development aid: bench.py ms_per_step / frac for the product and devlibs/librpsf_before.so, interleaved (argv: configs, e.g. "3 2 5")
for rep in 1 2 3; do
  for lib in devlibs/librpsf_before.so regularizepsf_amd/librpsf_hip.so; do
    for cfg in $1; do
      line=$(RPSF_LIB=$PWD/$lib timeout 300 python3 bench.py --config $cfg --no-cpu --no-e2e --steps 50 --warmup 5 2>/dev/null | tail -1)
      echo "$(basename $lib .so) rep=$rep config=$cfg $(echo "$line" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'new frames', d['roofline'].get('frac_new_frames'), 'parity', (d.get('parity') or {}).get('max_rel'))")"
    done
  done
done
