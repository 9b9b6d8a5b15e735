"""Step-by-step: the step kernels (debug outputs) against the oracle on the discrete ring of tests/test_gpu_rollout.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import binding as ob
from sustaingym_amd.hostio import to_host
from test_gpu_rollout import _gmm_engine, _moer_days, _ring
site, bins, N, bank, period = 'caltech', 5, 1022, 2048, 'Summer 2019'
debug = os.environ.get('DBG', '1') == '1'
net, eng = _gmm_engine(site, period, N, bank, seed=77, project=True, autoreset=True, debug_outputs=debug)
eng.set_autoreset_stride(N); eng.reset()
ring_t = _ring(eng, 'ringd', bins); ring = to_host(ring_t).copy()
ns, sess, req, day, _ = eng.download_episodes(0, bank)
bat = ob.OracleBatch(ob.OracleNetwork(net), N, 36, True)
bat.set_bank(ns, sess, req, day, _moer_days(site, period), autoreset_stride=N); bat.reset()
np.set_printoptions(linewidth=220, precision=6, suppress=True)
t = 0
for steps in (1, 95, 60, 132, 40, 300):
    for i in range(steps):
        if t >= 486:
            rem0, dep0, est0 = eng.station_state(); orem0, odep0, oest0 = bat.station_state(); obs_prev = o['obs'].copy()
        g = {k: to_host(v) for k, v in eng.step(ring_t[i % 5], bins=bins).items()}
        o = bat.step(ring[i % 5], bins=bins, autoreset=True, debug=True)
        t += 1
        bad = np.flatnonzero(np.abs(g['reward'] - o['reward']) > 1e-9)
        if len(bad):
            e = bad[0]
            print('t', t, 'episode step', t % 288, 'envs', bad[:5], 'reward', g['reward'][e], o['reward'][e])
            if debug:
                st = np.flatnonzero(g['pilots'][e] != o['pilots'][e])
                print(' stations', st, 'pilots gpu', g['pilots'][e][st], 'oracle', o['pilots'][e][st])
                print(' projected*32 gpu', g['projected'][e][st] * 32, 'oracle', o['projected'][e][st] * 32, 'action', ring[i % 5][e][st])
                occ = np.flatnonzero(o['pilots'][e] > 0)
                print(' all pilots oracle', o['pilots'][e][occ], 'gpu', g['pilots'][e][occ], 'stations', occ)
                print(' rem before: gpu', rem0[e][occ], 'oracle', orem0[e][occ], 'dep', dep0[e][occ], odep0[e][occ])
                print(' caps h = demand_f32/0.0173333:', (obs_prev[e][occ].astype(np.float64)) / (208/12000), 'status gpu', eng.env_scalars()['status'][e], 'oracle', o['status'][e])
                print(' slow count', eng.last_slow_count())
                allocc = np.flatnonzero(dep0[e] >= 0)
                print(' ALL plugged stations', allocc)
                print(' rem   ', rem0[e][allocc])
                print(' action', ring[i % 5][e][allocc])
                print(' h     ', np.minimum(32, obs_prev[e][allocc].astype(np.float64) / (208/12000)))
                print(' y gpu ', g['projected'][e][allocc] * 32)
                print(' y orc ', o['projected'][e][allocc] * 32)
            sys.exit(0)
print('no mismatch')
