#!/bin/bash
# Development aid: bench.py's ms_per_step for several builds of the library, interleaved and repeated.
#   scripts/sweep_libs.sh "<bench args>" lib1.so lib2.so ...
ARGS=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    ms=$(RPSF_LIB=$lib python3 bench.py --no-cpu $ARGS 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)
    echo "$(basename $lib .so) rep=$rep ms_per_step=$ms"
  done
done
