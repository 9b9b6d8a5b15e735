"""Fused rollout and loop of steps against the oracle's episode loop after T periods (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import binding as ob
from sustaingym_amd.hostio import to_host
from test_gpu_rollout import _gmm_engine, _run, _moer_days
site, policy, T = sys.argv[1], sys.argv[2], int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 1022
period = 'Summer 2019' if site == 'caltech' else 'Summer 2021'
for fused in (True, False):
    net, eng = _gmm_engine(site, period, N, 2048, seed=77, project=True, autoreset=False)
    eng.set_policy_seed(99, env_id_base=5000)
    obs0 = to_host(eng.reset()).copy()
    ns, sess, req, day, _ = eng.download_episodes(0, 2048)
    bat = ob.OracleBatch(ob.OracleNetwork(net), N, 36, True)
    bat.set_bank(ns, sess, req, day, _moer_days(site, period))
    assert np.array_equal(obs0, bat.reset())
    g = _run(eng, policy, T, 0, fused)
    o = bat.rollout(policy, obs0, steps=T, seed=99, env_id_base=5000)
    rem, dep, est = eng.station_state(); orem, odep, oest = bat.station_state()
    print('   status values', np.unique(eng.env_scalars()['status'], return_counts=True))
    bad = np.argwhere(np.abs(rem - orem) > 1e-9)
    print('fused' if fused else 'loop ', 'T', T, 'dep equal', np.array_equal(dep, odep), 'rem mismatches', len(bad),
          'max ret diff', np.abs(g['returns'] - o['returns']).max())
    if len(bad): print('   stations', np.unique(bad[:, 1], return_counts=True), 'wave in block', np.unique((bad[:, 0] // 4) % 4, return_counts=True))
    for e, s in bad[:6]:
        print('   env', e, 'st', s, rem[e, s], orem[e, s], (rem[e, s] - orem[e, s]) / 0.017333333)
