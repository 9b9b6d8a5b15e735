# Same-box A/B of bench.py configurations, interleaved repetitions.
# usage: python tools/ab_bench.py REPS "name1:ENV=V,ENV2=V;--flag ..." "name2:..." ...
#   each spec = name : comma-separated environment settings ; extra bench.py arguments
import json, os, subprocess, sys
import numpy as np
reps = int(sys.argv[1])
specs = []
for spec in sys.argv[2:]:
    name, rest = spec.split(':', 1)
    envs, _, args = rest.partition(';')
    env = dict(kv.split('=', 1) for kv in envs.split(',') if kv)
    specs.append((name, env, args.split()))
res = {}
for rep in range(reps):
    for name, env, args in specs:
        full = '/tmp/ab_bench_full.json'                    # the stdout line is the compact headline; the full record is here
        if os.path.exists(full):
            os.remove(full)
        subprocess.run([sys.executable, 'bench.py', '--no-secondary', '--no-cpu-baseline', '--full-out', full] + args,
                       env=dict(os.environ, **env), capture_output=True, text=True)
        if not os.path.exists(full):
            print(name, 'FAILED'); continue
        d = json.load(open(full))
        res.setdefault(name, []).append((d['ms_per_step'] * 1e3, d['roofline']['avg_kernel_ms'] * 1e3,
                                         d['roofline']['solver_kernel_ms'] * 1e3, d['roofline']['slow_queue_envs_per_step']))
for name, v in res.items():
    a = np.array(v)
    print(f'{name:<24} step us: min {a[:,0].min():.2f} med {np.median(a[:,0]):.2f} | kernel us: min {a[:,1].min():.2f} med {np.median(a[:,1]):.2f} | solver {np.median(a[:,2]):.2f} | slow envs {np.median(a[:,3]):.1f}')
