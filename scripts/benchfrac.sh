#!/bin/bash
# development: roofline.frac of bench configs for the library RPSF_LIB points at:  scripts/benchfrac.sh 6 7
for c in "$@"; do python bench.py --config $c --steps 50 --warmup 5 --no-cpu --no-e2e --new-frames 0 2>/dev/null | python3 -c "
import json,sys,os
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(os.environ.get('RPSF_LIB','product').split('/')[-1], 'config', $c, 'ms', l['ms_per_step'], 'frac', l['roofline']['frac'], 'parity', l.get('parity',{}).get('max_rel'))"; done
