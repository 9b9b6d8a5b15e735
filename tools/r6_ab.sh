# round 6: variants of the streaming kernel against the regular build, interleaved (GPU box): REPS=n tools/r6_ab.sh v1 v2 ...
# prints every run and, per arm, min / median of the pipelined step and of the one-launch step (us)
V=$PWD/sustaingym_amd/variants
R=${REPS:-3}
one() { python bench.py --no-secondary --no-cpu-baseline --full-out '' 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$1', r['ms_per_step'], ro['step_period_ms'], 'single', ro['single_launch']['ms_per_step'])"; }
T=$(mktemp)
for i in $(seq $R); do one base; for v in "$@"; do SUSTAINGYM_AMD_LIB=$V/lib_$v.so one $v; done; done | tee $T
python - $T <<'PY'
import sys, collections, statistics as st
d = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    p = ln.split()
    if len(p) == 5:
        d[p[0]].append((float(p[1]) * 1e3, float(p[4]) * 1e3))
for k, v in d.items():
    a, b = [x[0] for x in v], [x[1] for x in v]
    print(f'{k:<12} pipelined min {min(a):.2f} med {st.median(a):.2f} | single min {min(b):.2f} med {st.median(b):.2f}  (n={len(v)})')
PY
