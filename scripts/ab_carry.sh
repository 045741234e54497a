#!/bin/bash
# Development aid (run on the GPU box from the repo root): the carry timing experiment against the product build.
#   scripts/ab_carry.sh <outdir> lib1.so lib2.so ...
OUT=$1; shift
mkdir -p $OUT
{
IFS=';' read -ra CFGS <<< "${AB_CFGS:---config 3 --steps 50 --warmup 5;--config 3 --steps 40 --warmup 5 --rotate 8;--config 4 --steps 20 --warmup 3;--config 2 --steps 50 --warmup 5;--config 5 --steps 20 --warmup 3}"
for cfg in "${CFGS[@]}"; do
  echo "== bench.py $cfg"
  bash scripts/sweep_libs.sh "$cfg" "$@"
done
} > $OUT/ab.log 2>&1
cat $OUT/ab.log
REPO=$(pwd)
for lib in "$@"; do
  tag=$(basename $lib .so)
  export RPSF_LIB=$REPO/$lib
  bash scripts/pmc_passes.sh $OUT/pmc_$tag -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu > $OUT/pmc_$tag.log 2>&1
  grep -A40 "patch_kernel2_256p" $OUT/pmc_$tag/summary.txt | grep -E "FETCH_SIZE|WRITE_SIZE|TCC_HIT|TCC_MISS|RDREQ|WRREQ" | sed "s/^/$tag /"
  rm -rf $OUT/pmc_$tag/pass*
done
