#!/bin/bash
# tools/kernel_isa.sh lib.so 'mangled-name-regex' > out.s : disassembly of the matching gfx950 kernel(s)/function(s)
LIB=$1; PAT=$2
T=$(mktemp -d)
"$(dirname "$0")/unbundle.sh" "$LIB" $T
for co in $T/dev_*.co; do /opt/rocm/lib/llvm/bin/llvm-objdump -d --mcpu=gfx950 $co; done | awk -v pat="$PAT" '/^[0-9a-f]+ <.*>:$/ { on = ($0 ~ pat) } on { print }'
rm -rf $T
