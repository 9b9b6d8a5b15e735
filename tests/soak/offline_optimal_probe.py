"""Probe, not a test: can the second row recorded in the reference repository be reproduced?

examples/evcharging/env_validation.ipynb holds, as a comment, the reward breakdown of its OfflineOptimal
controller on caltech / 2020-02-01..05-31 / seed 2: reward 10.925368, profit 13.099406, carbon_cost 2.174038,
excess_charge 0.0, max_profit 14.45262.  max_profit is reproduced exactly (tests/test_event_generation.py).
This script restates the CURRENT OfflineOptimal formulation (algorithms/evcharging/baselines.py:130-223: LP
over the true sessions with the true MOER column, here solved with SciPy HiGHS) and realises its plan through
the oracle: reward 12.081, profit 13.877, carbon 1.796 with the legacy stepwise battery model, 12.023 / 13.810 / 1.787 with
acnportal's default continuous one (round 2) — neither is near the recorded row, whose carbon / profit ratio (0.166 vs 0.129)
says it charged at dirtier hours than any optimum of the current objective would.  The notebook cell predates the current code (its
other cells use the 4-tuple step / reset(return_info=True) API and a `reward_breakdown = oo.run(...)` that the
current BaseAlgorithm.run no longer returns), so the recorded row belongs to an older controller formulation;
it is NOT used as a pin."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
from scipy.optimize import linprog
from scipy.sparse import lil_matrix
from sustaingym_amd.event_generation import RealTraceGenerator
from sustaingym_amd.network import caltech_acn
from sustaingym_amd.envs import _pad_table, MAX_SESSIONS
from oracle import binding as ob

A_PERS_TO_KWH = (1 / 60) * (208 / 1000) * 5
PROFIT = A_PERS_TO_KWH * 0.15 * 0.20
CARBON = A_PERS_TO_KWH * 30.85 / 1000
print('factors', PROFIT, CARBON)
net = caltech_acn()
g = RealTraceGenerator('caltech', ('2020-02-01', '2020-05-31'))
g.set_seed(2)
table = g.get_event_table()
moer = g.get_moer()
E = len(table); T = 288; n = net.num_stations
arr = table.sessions['arrival'].astype(int); dep = table.sessions['departure'].astype(int); st = table.sessions['station'].astype(int)
print('sessions', E, 'max_profit', table.max_profit())
# variables
idx = {}
for e in range(E):
    for t in range(arr[e], min(dep[e], T)):
        idx[(e, t)] = len(idx)
nv = len(idx)
c = np.zeros(nv)
for (e, t), j in idx.items():
    c[j] = -(32 * PROFIT - 32 * CARBON * moer[t + 1, 0])
rows = []; rhs = []
A = lil_matrix((E + 16 * T, nv)); r = 0
for e in range(E):
    for t in range(arr[e], dep[e] - 1):
        if (e, t) in idx: A[r, idx[(e, t)]] = 1.0
    rhs.append(table.requested[e] / A_PERS_TO_KWH / 32); r += 1
M = np.asarray(net.constraint_matrix); ph = np.asarray(net.phase_angles); mag = np.asarray(net.magnitudes)
lin_rows = [k for k in range(M.shape[0]) if len(set(np.round(ph[M[k] != 0], 6))) == 1]
print('linear rows', lin_rows, 'of', M.shape[0])
by_t = {}
for (e, t), j in idx.items(): by_t.setdefault(t, []).append((e, j))
for t, lst in by_t.items():
    for k in lin_rows:
        any_ = False
        for e, j in lst:
            if M[k, st[e]] != 0:
                A[r, j] = abs(M[k, st[e]]) * 32; any_ = True
        if any_:
            rhs.append(mag[k]); r += 1
        else:
            A[r, :] = 0
A = A[:r].tocsr()
res = linprog(c, A_ub=A, b_ub=np.array(rhs), bounds=(0, 1), method='highs')
print('LP status', res.status, 'objective', -res.fun)
x = res.x
traj = np.zeros((n, T))
for (e, t), j in idx.items(): traj[st[e], t] = x[j]
# check cone rows
Z = (M * np.exp(1j * np.deg2rad(ph)))
viol = (np.abs(Z @ traj) * 32 - mag[:, None]).max(axis=1)
print('max violation per row (A):', np.round(viol, 3))
frac = ((x > 1e-6) & (x < 1 - 1e-6)).sum()
print('fractional entries', frac, 'of', nv, 'planned profit', 32 * PROFIT * x.sum())
# realise through the oracle
s, rq = _pad_table(table, MAX_SESSIONS)
bat = ob.OracleBatch(ob.OracleNetwork(net), 1, 36, True)
bat.set_bank(np.array([E], np.int32), s[None], rq[None], np.array([0], np.int32), moer[None])
bat.reset(np.array([0], np.int32))
tot = 0.0
for t in range(T):
    o = bat.step(traj[:, t][None].astype(np.float32))
    tot += float(o['reward'][0])
bd = o['breakdown'][0]
print(f'realised: reward {tot:.6f} profit {bd[0]:.6f} carbon {bd[1]:.6f} excess {bd[2]:.6f}   recorded: 10.925368 13.099406 2.174038 0.0')
