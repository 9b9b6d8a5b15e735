#!/bin/bash
# where the in-row water-filling's time goes: ablation builds (WRONG results, timing only) against the regular build on a
# synchronised Caltech GMM day by 4-hour block (p1 = one launch per step, p2 = pipelined halves)
#   pass0 / pass1: the Newton loop cut off after 0 / 1 passes (-DEVC_ABL_FILL_PASSES)   norev: no second evaluation of the rows
#   nofill: exact rows only (-DEVC_ABL_NO_FILL)
for v in base pass1 pass0 norev nofill base; do
  if [ $v = base ]; then echo "base $(python tools/gmm_days.py caltech 2>/dev/null)"
  else echo "$v $(SUSTAINGYM_AMD_LIB=$PWD/sustaingym_amd/variants/lib_$v.so python tools/gmm_days.py caltech 2>/dev/null)"; fi
done
