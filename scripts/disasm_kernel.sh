#!/bin/bash
# Development aid (no GPU needed): disassembly of one kernel of a built object.
#   scripts/disasm_kernel.sh regularizepsf_amd/build/product/k2_256p.o patch_kernel2_256p > /tmp/k.s
LLVM=${ROCM_PATH:-/opt/rocm}/lib/llvm/bin
tmp=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin "$1" $tmp/x.o 2>/dev/null
$LLVM/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/k.co
$LLVM/llvm-objdump -d $tmp/k.co | awk -v k="<$2>:" '$0 ~ k {f=1; print; next} f && /^[0-9a-f]+ <.*>:$/ {f=0} f'
rm -rf $tmp
