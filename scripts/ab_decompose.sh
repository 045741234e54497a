#!/bin/bash
# Development aid (GPU box, repo root): where the period of a persistent 256-pixel patch goes - ablation builds of patch_kernel2_256p
# (devlibs/abl_*.so, results wrong by design) against the product on bench.py's headline loop, interleaved.
#   scripts/ab_decompose.sh <outdir>
OUT=$1; mkdir -p $OUT
bash scripts/sweep_libs.sh "--steps 50 --warmup 5 --new-frames 0" regularizepsf_amd/librpsf_hip.so devlibs/abl_MEMONLY.so devlibs/abl_NOVALU.so \
  devlibs/abl_CHIPONLY.so devlibs/abl_NOMEM_NOLDS.so devlibs/abl_NOSTORE.so devlibs/abl_NOK.so devlibs/abl_NOGATHER.so devlibs/abl_NOLDS_NOBAR.so 2>&1 | tee $OUT/decompose.log
