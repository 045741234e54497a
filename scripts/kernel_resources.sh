#!/bin/bash
# Registers, spills, scratch, LDS and code size of the kernels in built objects (no GPU needed).
#   scripts/kernel_resources.sh regularizepsf_amd/build/product/k2_256p.o [more.o ...]
LLVM=${ROCM_PATH:-/opt/rocm}/lib/llvm/bin
for obj in "$@"; do
  tmp=$(mktemp -d)
  $LLVM/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $obj $tmp/x.o 2>/dev/null
  $LLVM/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/k.co
  echo "== $obj"
  $LLVM/llvm-readelf --notes $tmp/k.co | grep -E "\.name:|\.vgpr_count|\.agpr_count|\.sgpr_count|\.sgpr_spill_count|\.vgpr_spill_count|\.private_segment_fixed_size|\.group_segment_fixed_size" |
    awk '/\.name:/{if(n)print n": "l; n=$2; l=""; next}{gsub(/^ +\./,""); l=l" "$1$2}END{print n": "l}'
  $LLVM/llvm-readelf -s $tmp/k.co | awk '$4=="FUNC"{printf "  code bytes %s: %d\n", $8, $3}'
  rm -rf $tmp
done
