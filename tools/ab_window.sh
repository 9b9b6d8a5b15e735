#!/bin/bash
# tools/ab_window.sh [reps]: the driver's 20-step window (bench.py --steps 20 --warmup 5), interleaved: plain; the second train
# started EVC_PIPE_SKEW_US late at a cold start (a one-wavefront sleep kernel; a host gap > EVC_PIPE_GAP_US counts as drained);
# the phased start by construction (EVC_PIPE_PHASE=1).
R=${1:-4}
one() { python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['ms_per_step'])"; }
for i in $(seq $R); do one plain; EVC_PIPE_SKEW_US=11 EVC_PIPE_GAP_US=60 one skew11; EVC_PIPE_SKEW_US=7 EVC_PIPE_GAP_US=60 one skew7; done
