#!/bin/bash
# Development aid (GPU box, repo root): the split-patch timing skeleton (devlibs/split*.so, -DRPSF_DEV_SPLIT, results wrong by design)
# against the product on bench.py's headline loop.   scripts/ab_split.sh <outdir> lib1.so lib2.so ...
OUT=$1; shift; mkdir -p $OUT
first=$1
# a small frame first, under a timeout: a protocol slip in a skeleton must not hang the box
RPSF_LIB=$first timeout 120 python3 scripts/kbench.py --n 256 --size 1024 --iters 5 --tag small > $OUT/small.log 2>&1 || { echo "small run failed"; cat $OUT/small.log; exit 1; }
cat $OUT/small.log
bash scripts/sweep_libs.sh "--steps 50 --warmup 5 --new-frames 0" regularizepsf_amd/librpsf_hip.so "$@" 2>&1 | tee $OUT/split.log
