#!/bin/bash
# tools/ab_window.sh [reps]: the driver's 20-step window (bench.py --steps 20 --warmup 5) with the phased cold start of the
# pipelined trains on / off (EVC_PIPE_PHASE; EVC_PIPE_GAP_US = host gap that counts as "drained"), interleaved.
R=${1:-4}
one() { python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['ms_per_step'])"; }
for i in $(seq $R); do EVC_PIPE_PHASE=1 EVC_PIPE_GAP_US=40 one phased40; EVC_PIPE_PHASE=0 one plain; EVC_PIPE_PHASE=0 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --pipeline 1 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(\"single\", r[\"value\"], r[\"ms_per_step\"])"; done
