#!/bin/bash
for rep in 1 2; do for v in base scan; do
  export SUSTAINGYM_AMD_LIB=$(pwd)/sustaingym_amd/variants/lib_$v.so
  echo "$v: $(tools/scratch/trace_gmm.sh caltech 2>&1 | tail -1)"
done; done
