#!/bin/bash
# Development aid (needs the -DRPSF_DEV_ENV build): one host frame end to end under the knobs of host_one_frame_banded
export RPSF_LIB=$PWD/regularizepsf_amd/librpsf_hip_dev.so
for knobs in "RPSF_X=1" "RPSF_FRAME_EVERY=2" "RPSF_FRAME_EVERY=3" "RPSF_FRAME_EVERY=4" "RPSF_HOST_THREADS=32 RPSF_FRAME_EVERY=4" "RPSF_HOST_THREADS=24 RPSF_FRAME_EVERY=3" "RPSF_HOST_THREADS=8"; do
  echo "== $knobs"
  env $knobs timeout 200 python scripts/host_frame_bands.py 2>&1 | grep -E "bands  (4|6|8)"
done
