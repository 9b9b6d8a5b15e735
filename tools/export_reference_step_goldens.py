#!/usr/bin/env python
"""Records step-level golden vectors from the REFERENCE — run once wherever `sustaingym` (chrisyeh96/sustaingym),
acnportal and gymnasium import (plus cvxpy + MOSEK for the projection-on cases); never shipped to or run on the GPU box.

    python tools/export_reference_step_goldens.py [--out tests/golden/reference_steps.npz] [--steps 288]
                                                  [--seeds 0 1 2] [--no-projection-on] [--networks-dir sustaingym_amd/data]

What it pins (VERDICT r2 "missing" #2): EVChargingEnv.step (env.py:229-291) end to end — acnportal's
simulator, battery model, event ordering, EVSE rate sets, the observation, the reward and, with --projection-on, the
cvxpy/MOSEK projection — for >= 3 seeds x {caltech, jpl} x {projection off, on} x {continuous, DiscreteActionWrapper(5)}.
Per case it stores the episode the reference generator produced (session table + MOER matrix), the actions it was
stepped with, and per step: the flattened observation (spaces.flatten key order), reward, terminated, the schedule sent to
the simulator (pilots), the delivered rates, and the reward breakdown.  tests/test_reference_steps.py replays the file
through the oracle (CPU) and through the HIP engine (-m gpu).

The same run exports the two charging networks exactly as the reference reads them (utils.py:83-88 -> acnportal):
`<networks-dir>/caltech_acn.json`, `<networks-dir>/jpl_acn.json` (ChargingNetwork.from_acnportal) — the JPL file replaces
the provisional built-in constraint set (sustaingym_amd/network.py).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PERIOD = ('2021-05-01', '2021-08-31')         # "Summer 2021" of the reference's DEFAULT_PERIOD_TO_RANGE (utils.py:48-64)
KEYS = ('demands', 'est_departures', 'forecasted_moer', 'prev_moer', 'timestep')     # gymnasium.spaces.flatten order


def flatten(obs: dict) -> np.ndarray:
    return np.concatenate([np.asarray(obs[k], dtype=np.float32).ravel() for k in KEYS])


def session_table(evs, station_ids) -> dict:
    idx = {s: i for i, s in enumerate(station_ids)}
    order = sorted(range(len(evs)), key=lambda i: evs[i].arrival)            # stable: the event queue's plug-in order
    return {
        'arrival': np.array([evs[i].arrival for i in order], np.int16),
        'departure': np.array([evs[i].departure for i in order], np.int16),
        'est_departure': np.array([evs[i].estimated_departure for i in order], np.int16),
        'station': np.array([idx[evs[i].station_id] for i in order], np.int16),
        'requested': np.array([evs[i].requested_energy for i in order], np.float64),
    }


def run_case(site, generator_kind, seed, project, bins, steps):
    from sustaingym.envs.evcharging import EVChargingEnv, GMMsTraceGenerator, RealTraceGenerator
    try:
        from sustaingym.envs.evcharging import DiscreteActionWrapper
    except ImportError:                                   # the reference's __init__ imports a module that does not exist
        from sustaingym.envs.wrappers import DiscreteActionWrapper
    gen = (RealTraceGenerator if generator_kind == 'real' else GMMsTraceGenerator)(site, PERIOD)
    base = EVChargingEnv(gen, project_action_in_env=project)
    env = DiscreteActionWrapper(base, bins=bins) if bins else base
    obs, info = env.reset(seed=seed)
    n = base.num_stations
    rng = np.random.default_rng(1000 + seed)
    rec = {'obs0': flatten(obs), 'max_profit': float(info['max_profit']), 'moer': np.asarray(base.moer, np.float64),
           **{'table_' + k: v for k, v in session_table(base._evs, base.cn.station_ids).items()}}
    actions, obs_l, rew, term, pilots, rates, bd = [], [], [], [], [], [], []
    for t in range(steps):
        a = rng.integers(0, bins, n).astype(np.int64) if bins else rng.random(n, dtype=np.float32)
        if t % 7 == 3 and not bins:
            a[:] = 1.0                                         # saturated steps: pods and feeders bind, ties of env.py:373-378
        obs, r, terminated, truncated, info = env.step(a.copy())          # (env.py:366 scales the caller's array in place)
        sim = base._simulator
        actions.append(a)
        obs_l.append(flatten(obs))
        rew.append(float(r))
        term.append(bool(terminated))
        pilots.append(np.asarray(sim.pilot_signals[:, base.t - 1], np.float64))
        rates.append(np.asarray(sim.charging_rates[:, base.t - 1], np.float64))
        b = info['reward_breakdown']
        bd.append([b['profit'], b['carbon_cost'], b['excess_charge']])
        if terminated:
            break
    rec.update(actions=np.array(actions), obs=np.array(obs_l), reward=np.array(rew), terminated=np.array(term),
               pilots=np.array(pilots), rates=np.array(rates), breakdown=np.array(bd, np.float64))
    env.close()
    return rec


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden', 'reference_steps.npz'))
    ap.add_argument('--steps', type=int, default=288)
    ap.add_argument('--seeds', type=int, nargs='+', default=[0, 1, 2])
    ap.add_argument('--no-projection-on', action='store_true', help='skip the cases that need cvxpy + MOSEK')
    ap.add_argument('--networks-dir', default=os.path.join(ROOT, 'sustaingym_amd', 'data'))
    args = ap.parse_args()

    import acnportal                                           # noqa: F401  (fail early and clearly where the reference cannot run)
    from acnportal.acnsim.network.sites import caltech_acn, jpl_acn
    from sustaingym_amd.network import ChargingNetwork
    for site, factory in (('caltech', caltech_acn), ('jpl', jpl_acn)):
        path = os.path.join(args.networks_dir, f'{site}_acn.json')
        ChargingNetwork.from_acnportal(factory(), site).to_json(path)
        print('wrote', path)

    out, cases = {}, []
    for site in ('caltech', 'jpl'):
        for project in ([False] if args.no_projection_on else [False, True]):
            for bins in (0, 5):
                for seed in args.seeds:
                    kind = 'real' if seed % 2 == 0 else 'gmm'
                    name = f'{site}|{kind}|seed{seed}|project{int(project)}|bins{bins}'
                    rec = run_case(site, kind, seed, project, bins, args.steps)
                    for k, v in rec.items():
                        out[f'{name}|{k}'] = v
                    cases.append(name)
                    print(name, 'steps', len(rec['reward']), 'return', float(rec['reward'].sum()))
    import acnportal as _a
    out['cases'] = np.array(cases)
    out['versions'] = np.array([f'acnportal {getattr(_a, "__version__", "?")}', f'numpy {np.__version__}'])
    np.savez_compressed(args.out, **out)
    print('wrote', args.out, f'({len(cases)} cases)')


if __name__ == '__main__':
    main()
