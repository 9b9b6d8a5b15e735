"""Finds the first period at which the fused rollout and the loop of steps diverge (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from test_gpu_rollout import _gmm_engine, _run
site, policy = sys.argv[1], sys.argv[2]
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 1
N, bank = 1022, 2048
period = 'Summer 2019' if site == 'caltech' else 'Summer 2021'
engs = []
for fused in (True, False):
    net, eng = _gmm_engine(site, period, N, bank, seed=77, project=True, autoreset=True)
    eng.set_autoreset_stride(N); eng.set_policy_seed(99, env_id_base=5000); eng.reset(); engs.append(eng)
t = 0
while t < 288:
    a = _run(engs[0], policy, chunk, 0, True); b = _run(engs[1], policy, chunk, 0, False); t += chunk
    sa, sb = engs[0].get_state(), engs[1].get_state()
    bad = np.argwhere(sa['remaining_kwh'] != sb['remaining_kwh'])
    sc = np.argwhere(sa['scalars'] != sb['scalars'])
    dd = np.argwhere(sa['departure'] != sb['departure'])
    if len(bad) or len(sc) or len(dd):
        print('t', t, 'rem mismatches', len(bad), 'scalar', len(sc), 'dep', len(dd))
        for e, s in bad[:10]:
            print(' env', e, 'st', s, sa['remaining_kwh'][e, s], sb['remaining_kwh'][e, s], sa['remaining_kwh'][e, s] - sb['remaining_kwh'][e, s])
        e = bad[0][0] if len(bad) else (sc[0][0] if len(sc) else dd[0][0])
        print(' env', e, 'scalars', sa['scalars'][e], sb['scalars'][e])
        print(' rem fused', sa['remaining_kwh'][e][sa['departure'][e] >= 0]); print(' rem loop ', sb['remaining_kwh'][e][sb['departure'][e] >= 0])
        print(' slow count loop', engs[1].last_slow_count())
        break
else:
    print('no divergence')
