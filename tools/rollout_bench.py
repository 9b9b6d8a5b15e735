"""tools/rollout_bench.py [N] — env-steps/s of whole-episode rollouts under the device-resident policies: the fused
rollout kernel (one launch per episode, csrc/evc_rollout.h) beside the loop of evc_step launches (EVC_ROLLOUT_FUSED=0),
on bench.py's synthetic days and on device-generated GMM days.  One JSON line per configuration."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import EvWorkload  # noqa: E402


def run(site, episodes, policy, fused, N, reps=3, project=True):
    w = EvWorkload(site, N, 0, 0, project=project, episodes=episodes, phase='sync')
    eng = w.eng
    eng.set_policy_seed(7)
    os.environ['EVC_ROLLOUT_FUSED'] = '1' if fused else '0'
    times = []
    if fused:                               # synchronised calls: what the engine needs to settle on a register budget (launch_rollout)
        for r in range(8):
            eng.rollout(policy=policy, steps=288)
            torch.cuda.synchronize()
    for r in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.rollout(policy=policy, steps=288)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    os.environ.pop('EVC_ROLLOUT_FUSED')
    best = min(times[1:])
    rec = {'site': site, 'episodes': episodes, 'policy': policy, 'fused': fused, 'N': N, 'project': project,
           'episode_ms': round(best * 1e3, 3), 'us_per_step': round(best / 288 * 1e6, 2),
           'env_steps_per_s': round(N * 288 / best, 1), 'all_ms': [round(t * 1e3, 2) for t in times],
           'waves_per_simd': eng.last_rollout_waves() if fused else None}
    w.close()
    return rec


if __name__ == '__main__':
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    only = sys.argv[2] if len(sys.argv) > 2 else ''
    for site, episodes in (('caltech', 'synthetic'), ('caltech', 'gmm'), ('jpl', 'gmm')):
        for policy in ('greedy', 'random'):
            for fused in (True, False):
                if only == 'fused' and not fused:
                    continue
                print(json.dumps(run(site, episodes, policy, fused, N)), flush=True)
