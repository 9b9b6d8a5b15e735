#!/bin/bash
# tools/kernel_isa.sh lib.so 'mangled-name-regex' > out.s : disassembly of the matching gfx950 kernel(s)/function(s)
LIB=$1; PAT=$2
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$LIB" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
/opt/rocm/lib/llvm/bin/llvm-objdump -d --mcpu=gfx950 $T/dev.co | awk -v pat="$PAT" '/^[0-9a-f]+ <.*>:$/ { on = ($0 ~ pat) } on { print }'
rm -rf $T
