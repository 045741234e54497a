#!/bin/bash
# Development aid (no GPU needed): one hash per kernel object of the product build, over the instruction text of its gfx950 code object -
# two source trees that print the same hashes compile to the same device code.   scripts/code_object_hashes.sh [objdir]
LLVM=${ROCM_PATH:-/opt/rocm}/lib/llvm/bin
dir=${1:-regularizepsf_amd/build/product}
for o in "$dir"/*.o; do
  tmp=$(mktemp -d)
  $LLVM/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin "$o" $tmp/x.o 2>/dev/null
  $LLVM/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/k.co 2>/dev/null
  h=$($LLVM/llvm-objdump -d $tmp/k.co 2>/dev/null | grep -v "file format" | md5sum | cut -c1-16)
  echo "$(basename $o) $h"
  rm -rf $tmp
done
