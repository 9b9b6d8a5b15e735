"""One (or a few) fused rollouts of one configuration, for profiling: site episodes policy [N] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import EvWorkload
site, episodes, policy = sys.argv[1:4]
N = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
w = EvWorkload(site, N, 0, 0, project=True, episodes=episodes, phase='sync')
w.eng.set_policy_seed(7)
for _ in range(reps):
    w.eng.rollout(policy=policy, steps=288)
torch.cuda.synchronize()
w.close()
