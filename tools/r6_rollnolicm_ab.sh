# -mllvm -disable-machine-licm for the rollout translation unit against the regular build (GPU box): both register budgets,
# synthetic + GMM days, greedy + random (tools/r6_rollout_waves.py), three rounds each.
V=$PWD/sustaingym_amd/variants/lib_rollnolicm.so
for round in 1 2 3; do
for lib in base rollnolicm; do
  if [ $lib = rollnolicm ]; then export SUSTAINGYM_AMD_LIB=$V; else unset SUSTAINGYM_AMD_LIB; fi
  for w in 2 3; do
    echo "== $lib waves=$w round $round"; EVC_ROLLOUT_WAVES=$w python tools/r6_rollout_waves.py 2>/dev/null | cut -c1-200
  done
done
done
