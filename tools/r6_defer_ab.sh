# round 6: wide quads deferred to a second pass (regular build) against every quad stepped inside the loop (-DEVC_DEFER_WIDE=0), GPU box
V=$PWD/sustaingym_amd/variants/lib_nodefer.so
REPS=5 bash tools/r6_ab.sh nodefer | tail -2
for lib in base nodefer; do
  if [ $lib = nodefer ]; then export SUSTAINGYM_AMD_LIB=$V; else unset SUSTAINGYM_AMD_LIB; fi
  echo "== $lib: no projection"; for i in 1 2 3; do python bench.py --no-project --no-secondary --no-cpu-baseline --full-out '' 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['single_launch']['ms_per_step'])"; done
  echo "== $lib: gmm days"; python tools/gmm_days.py 2>/dev/null | cut -c1-400
  echo "== $lib: gmm days again"; python tools/gmm_days.py 2>/dev/null | cut -c1-400
  echo "== $lib: SQ"; tools/pmc_sq.sh 2>/dev/null | tail -1 | cut -c1-330
done
