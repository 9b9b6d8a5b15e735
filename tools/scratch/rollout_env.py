"""What the step kernels do with one environment at one period (debug aid): projected actions / pilots."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from test_gpu_rollout import _gmm_engine, _run
site, policy, T, env = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
period = 'Summer 2019' if site == 'caltech' else 'Summer 2021'
net, eng = _gmm_engine(site, period, 1022, 2048, seed=77, project=True, autoreset=False, debug_outputs=True)
eng.set_policy_seed(99, env_id_base=5000); eng.reset()
_run(eng, policy, T - 1, 0, False)
rem, dep, est = eng.station_state()
g = eng.step_policy(policy)
occ = dep[env] >= 0
np.set_printoptions(linewidth=200, precision=6, suppress=True)
print('stations', np.flatnonzero(occ)); print('rem      ', rem[env][occ]); print('projected*32', g['projected'][env][occ] * 32); print('pilots   ', g['pilots'][env][occ])
print('slow count', eng.last_slow_count())
print('quad mates occupancy', [(int((dep[e] >= 0).sum())) for e in range(env // 4 * 4, env // 4 * 4 + 4)])
