# Interleaved A/B of library variants (each in its own subprocess would reload HIP; instead run
# sequentially per lib but repeat the whole sequence): prints min/median us per step.
import subprocess, sys, os, json, numpy as np
libs = sys.argv[1].split(',')
script = sys.argv[2] if len(sys.argv) > 2 else 'tools/ab_caps.py'      # e.g. tools/ab_kernel.py
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = {}
for rep in range(reps):
    for lib in libs:
        env = dict(os.environ, SUSTAINGYM_AMD_LIB=os.path.join(os.getcwd(), 'sustaingym_amd/variants', f'lib_{lib}.so'))
        out = subprocess.run([sys.executable, script, '1024'], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if line.startswith('project='):
                key = (lib, line.split()[0])
                res.setdefault(key, []).append(float(line.split('min ')[1].split()[0]))
for key, v in sorted(res.items()):
    print(key, 'min', min(v), 'median', float(np.median(v)), v)
