"""Replay of a discrete action ring: fused rollout and loop of steps against the oracle (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import binding as ob
from sustaingym_amd.hostio import to_host
from test_gpu_rollout import _gmm_engine, _run, _moer_days, _ring
site, policy, bins = 'caltech', 'ringd', 5
N, bank = 1022, 2048
period = 'Summer 2019'
res = {}
for fused in (True, False):
    net, eng = _gmm_engine(site, period, N, bank, seed=77, project=True, autoreset=True)
    eng.set_autoreset_stride(N); eng.reset()
    ring = to_host(_ring(eng, policy, bins)).copy()
    rets = []
    for steps in (1, 95, 60, 132, 40, 300):
        rets.append(_run(eng, policy, steps, bins, fused)['returns'])
    res[fused] = rets
    if fused:
        ns, sess, req, day, _ = eng.download_episodes(0, bank)
    eng.close()
bat = ob.OracleBatch(ob.OracleNetwork(net), N, 36, True)
bat.set_bank(ns, sess, req, day, _moer_days(site, period), autoreset_stride=N)
bat.reset()
for ci, steps in enumerate((1, 95, 60, 132, 40, 300)):
    ret = np.zeros(N)
    for i in range(steps):
        ret += bat.step(ring[i % 5], bins=bins, autoreset=True, debug=False)['reward']
    for fused in (True, False):
        d = np.abs(res[fused][ci] - ret)
        bad = np.flatnonzero(d > 1e-9 * np.maximum(1, np.abs(ret)))
        print('chunk', ci, 'fused' if fused else 'loop ', 'mismatching envs vs oracle:', bad[:5], d[bad][:5])
