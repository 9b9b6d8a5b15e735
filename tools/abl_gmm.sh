#!/bin/bash
# tools/abl_gmm.sh NAME...: the Caltech GMM day by 4-hour block for the regular library and ablation variants (WRONG results: timing only)
echo "regular $(python tools/gmm_days.py caltech 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['single_launch']['ms_per_step'], r['kernel_us_by_4h'])")"
for n in "$@"; do
  echo "$n $(SUSTAINGYM_AMD_LIB=$PWD/sustaingym_amd/variants/lib_$n.so python tools/gmm_days.py caltech 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['single_launch']['ms_per_step'], r['kernel_us_by_4h'])")"
done
