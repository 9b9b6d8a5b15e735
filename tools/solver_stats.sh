#!/bin/bash
# tools/solver_stats.sh [out.txt]: slow-path work statistics per queued environment, both sites (GPU box, repo root).
# Needs sustaingym_amd/variants/lib_stats.so = the CURRENT sources with -DEVC_SOLVER_STATS=1 (built in the build container:
# `tools/build_variant.sh stats "-DEVC_SOLVER_STATS=1"`; variants/ travels with gpurun).  Refuses a variant that is older than
# the regular library (round 4 committed a Python traceback from a stale one as evidence).
OUT=${1:-gpurun_out/solver_stats.txt}
V=sustaingym_amd/variants/lib_stats.so
if [ ! -f $V ] || [ $V -ot sustaingym_amd/libevcharge_hip.so ]; then
  echo "solver_stats.sh: $V is missing or older than the regular library: rebuild it (tools/build_variant.sh stats \"-DEVC_SOLVER_STATS=1\")" >&2
  exit 2
fi
{ echo "# tools/solver_stats.sh on $(date -u +%F) — library $(sha256sum sustaingym_amd/libevcharge_hip.so | cut -c1-16), stats variant $(sha256sum $V | cut -c1-16)"
  python tools/solver_stats.py caltech 2>/dev/null
  python tools/solver_stats.py jpl 2>/dev/null; } | tee $OUT
grep -q "cycles/env" $OUT || { echo "solver_stats.sh: no statistics in $OUT" >&2; exit 3; }
