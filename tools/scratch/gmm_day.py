"""One synchronised GMM (or real / synthetic) day at 65 536 environments: wall and kernel time per step by 4-hour block.
Usage: gmm_day.py site [episodes]   (env: EVC_DRAIN, EVC_DRAIN_MAXQ ...)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
site = sys.argv[1]
episodes = sys.argv[2] if len(sys.argv) > 2 else 'gmm'
r = bench.secondary_days(site, episodes, 0, 'continuous')
r.pop('workload'); r.pop('roofline')
print(json.dumps({'site': site, 'episodes': episodes, 'EVC_DRAIN': os.environ.get('EVC_DRAIN'), 'MAXQ': os.environ.get('EVC_DRAIN_MAXQ'), **r}))
