import sys, json
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import rollout_bench as rb
for site in ('caltech', 'jpl'):
    for project in (True, False):
        r = rb.run(site, 'gmm', 'greedy', True, 65536, project=project)
        print(json.dumps({k: r[k] for k in ('site', 'project', 'us_per_step', 'env_steps_per_s', 'waves_per_simd')}), flush=True)
