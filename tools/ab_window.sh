#!/bin/bash
# tools/ab_window.sh [reps]: the driver's 20-step window (bench.py --steps 20 --warmup 5) with the phased cold start of the
# pipelined trains on / off (EVC_PIPE_PHASE), interleaved; then the default long run.
R=${1:-4}
one() { python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['ms_per_step'])"; }
for i in $(seq $R); do EVC_PIPE_PHASE=1 one phased; EVC_PIPE_PHASE=0 one plain; done
python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('default', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline'].get('step_period_ms'))"
EVC_PIPE_PHASE=0 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('default-nophase', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline'].get('step_period_ms'))"
