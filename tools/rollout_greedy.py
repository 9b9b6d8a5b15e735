import sys, json
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import rollout_bench as rb
for site in (sys.argv[1:] or ['caltech', 'jpl']):
    for policy in ('greedy', 'random'):
        r = rb.run(site, 'gmm', policy, True, 65536)
        print(json.dumps({k: r[k] for k in ('site', 'policy', 'us_per_step', 'env_steps_per_s', 'waves_per_simd')}), flush=True)
