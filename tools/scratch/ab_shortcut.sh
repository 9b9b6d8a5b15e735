#!/bin/bash
# same-box A/B: variants/lib_base.so against the built library — headline (pipelined), GMM days by block, GPU suite
python tools/ab_bench.py 4 "base:SUSTAINGYM_AMD_LIB=sustaingym_amd/variants/lib_base.so;--no-single-launch" "new:;--no-single-launch" 2>&1 | tail -2
for s in caltech jpl; do
  SUSTAINGYM_AMD_LIB=sustaingym_amd/variants/lib_base.so python tools/scratch/gmm_blocks.py $s 2>&1 | tail -1
  python tools/scratch/gmm_blocks.py $s 2>&1 | tail -1
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
