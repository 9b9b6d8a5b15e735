#!/bin/bash
# tools/ab_variant.sh NAME [reps]: the regular library against sustaingym_amd/variants/lib_NAME.so, interleaved: bench.py's
# headline (pipelined and one launch per step) and the GMM days.
V=$PWD/sustaingym_amd/variants/lib_$1.so; R=${2:-3}
one() { python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$1', r['ms_per_step'], 'single', (ro.get('single_launch') or {}).get('ms_per_step'))"; }
for i in $(seq $R); do one base; SUSTAINGYM_AMD_LIB=$V one $1; done
echo base gmm; python tools/gmm_days.py 2>/dev/null | cut -c1-70
echo $1 gmm; SUSTAINGYM_AMD_LIB=$V python tools/gmm_days.py 2>/dev/null | cut -c1-70
