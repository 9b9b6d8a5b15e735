#!/bin/bash
# same-box A/B of a water-filling variant (sustaingym_amd/variants/lib_$1.so, tools/build_variant.sh) against the regular
# build: the headline step (ms per step, step period, half launch, single launch), then a synchronised GMM day by 4-hour
# block (p1 = one launch per step, p2 = pipelined halves; last figure of each list = the day's mean)
VN=${1:-warm32}
V=$PWD/sustaingym_amd/variants/lib_$VN.so
hl() { python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], r.get('step_period_ms'), r.get('half_launch_ms'), r['single_launch']['ms_per_step'])"; }
for rep in 1 2; do
  echo "$VN headline $(SUSTAINGYM_AMD_LIB=$V hl)"
  echo "base headline $(hl)"
  for site in caltech jpl; do
    echo "$VN $(SUSTAINGYM_AMD_LIB=$V python tools/scratch/gmm_blocks.py $site 2>/dev/null)"
    echo "base $(python tools/scratch/gmm_blocks.py $site 2>/dev/null)"
  done
done
