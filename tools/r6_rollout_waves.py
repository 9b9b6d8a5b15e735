import sys, json
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import rollout_bench as rb
for episodes in ('synthetic', 'gmm'):
    for policy in ('greedy', 'random'):
        r = rb.run('caltech', episodes, policy, True, 65536)
        print(json.dumps({'episodes': episodes, **{k: r[k] for k in ('policy', 'us_per_step', 'env_steps_per_s', 'waves_per_simd')}}), flush=True)
