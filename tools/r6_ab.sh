# round 6: variants of the streaming kernel against the regular build, interleaved (GPU box)
mkdir -p gpurun_out/r6e
V=$PWD/sustaingym_amd/variants
one() { python bench.py --no-secondary --no-cpu-baseline --full-out '' 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$1', r['ms_per_step'], ro['step_period_ms'], 'single', ro['single_launch']['ms_per_step'])"; }
for i in 1 2 3; do one base; for v in "$@"; do SUSTAINGYM_AMD_LIB=$V/lib_$v.so one $v; done; done
