#!/bin/bash
# Development aid: which side of a host frame slows the copy engines down - staging, widening, or their threads (argv: output directory).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$1; mkdir -p $O
for threads in 16 8 4; do
  for cfg in "8 0 1 0" "8 1 0 0" "8 1 0 1" "8 0 0 1"; do
    tag=t${threads}_$(echo $cfg | tr " " _)
    RPSF_HOST_THREADS=$threads timeout 250 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl_$tag -- python3 $R/scripts/host_frame_timeline.py 4096 256 $cfg > $O/tl_$tag.log 2>&1
    python3 $R/scripts/host_frame_timeline_read.py $O/tl_$tag > $O/tl_$tag.txt 2>&1
    rm -rf $O/tl_$tag
    echo "threads $threads bands/pin_in/pin_out/f64 = $cfg: $(grep 'apply' $O/tl_$tag.log | sort -k3 -n | head -1)"
    tail -3 $O/tl_$tag.txt
  done
done
