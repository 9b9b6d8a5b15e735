import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import rollout_bench as rb
site, episodes, policy = sys.argv[1:4]
for fused in (True, False):
    print(json.dumps({k: v for k, v in rb.run(site, episodes, policy, fused, 65536, reps=2).items() if k in ('site', 'episodes', 'policy', 'fused', 'episode_ms', 'env_steps_per_s', 'waves_per_simd')}), flush=True)
