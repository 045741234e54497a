#!/bin/bash
# Development aid: bench.py's ms_per_step AND ms_per_step_new_frames for several builds of the library, interleaved and repeated.
#   scripts/sweep_libs_nf.sh "<bench args>" lib1.so lib2.so ...
ARGS=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    out=$(RPSF_LIB=$lib python bench.py --no-cpu $ARGS 2>/dev/null)
    ms=$(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)
    nf=$(echo "$out" | grep -o '"ms_per_step_new_frames": [0-9.]*' | cut -d' ' -f2)
    pf=$(echo "$out" | grep -o '"ms_per_step_new_frames_with_image_prefetch": [0-9.]*' | cut -d' ' -f2)
    echo "$(basename $lib .so) rep=$rep ms_per_step=$ms new_frames=$nf new_frames_with_opt_in_prefetch=$pf"
  done
done
