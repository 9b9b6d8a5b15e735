#!/usr/bin/env python
"""Generates tests/golden/step_units.npz by RUNNING THE REFERENCE's own step code — the Python half of
``EVChargingEnv.step`` — in this container (VERDICT r3 "next" #1).  Container only: imports /root/reference through
tools/ref_import.py (``install_step_stubs``: base classes and containers are stubbed, no arithmetic; the stubs say
which), never travels to the GPU box; what travels is the .npz (data: inputs and the reference's outputs).

    python tests/golden/make_step_unit_golden.py

What runs UNMODIFIED from /root/reference, as bound methods of real instances:
  EVChargingEnv.__init__ (project_action_in_env=False)          env.py:116-176   spaces, buffers, key order
  EVChargingEnv class constants                                  env.py:99-114
  EVChargingEnv._to_schedule                                     env.py:340-379   a*32, AV / CC legal pilots, np.round
  EVChargingEnv._get_observation                                 env.py:381-394
  EVChargingEnv._get_reward                                      env.py:431-464   incl. the cumulative breakdown
  DiscreteActionWrapper.action                                   wrappers.py:43-45
  MultiAgentEVChargingEnv._create_dict_from_obs_agg              multiagent_env.py:102-148 (periods_delay 0 and 3)

What the reference delegates to packages that are NOT in this image is supplied as FIXTURE INPUT, recorded in the file:
  * ``cn.station_ids`` / ``cn.min_pilot_signals``: the packaged network descriptor (sustaingym_amd/network.py);
  * ``self._interface.active_sessions()`` (acnportal): the (station, estimated_departure, remaining_demand) records of
    the episode's plugged-in, not fully charged EVs — taken from the ORACLE's state after the same step;
  * ``self._simulator.charging_rates[:, t-1]`` (acnportal's battery model): the oracle's delivered amps of that step;
  * ``network.constraint_current(schedule)`` (acnportal): ``A_tilde @ schedule`` with ``A_tilde`` formed here the way
    env.py:485-486 forms it (``constraint_matrix * exp(1j * deg2rad(_phase_angles))``) from the REFERENCE's own
    schedule of that step; ``network.magnitudes``: the packaged descriptor;
  * with projection (cvxpy + MOSEK cannot run): the oracle's projected action is handed to ``_to_schedule`` of an
    env built with projection off, so the scaling and rounding of PROJECTED float64 values is reference-run too.
So the file pins, against the reference itself: a1, a2, and the reference-side halves of a8 / a9 / a14.
It does NOT pin acnportal's simulator (a4-a7) or the projection (a3): those inputs come from the oracle.

Cases: {caltech, jpl} x {real day + continuous, GMM day + DiscreteActionWrapper(5), GMM day + projection on,
rounding boundaries}.  tests/test_step_units.py replays them through the oracle (CPU) and the HIP engine (-m gpu).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, ROOT)
from ref_import import _Bag, reference_generators, reference_step_modules  # noqa: E402

from oracle import binding as ob  # noqa: E402  (tests/ may use the oracle: it is the checker)
from sustaingym_amd.network import site_str_to_site  # noqa: E402

FULLY_CHARGED_EPS = 1e-3   # acnportal EV.fully_charged [MEM]; decides which records are INPUT, no reference code tests it


def table_of(df, station_ids, cap=100):
    idx = {s: i for i, s in enumerate(station_ids)}
    arr = df['arrival'].values.astype(np.int16)
    dep = df['departure'].values.astype(np.int16)
    est = df['estimated_departure'].values.astype(np.int16)
    st = np.array([idx[s] for s in df['station_id']], dtype=np.int16)
    req = np.minimum(df['requested_energy (kWh)'].values.astype(np.float64), cap)
    return arr, dep, est, st, req


def boundary_amps():
    """Pilot-signal values (amps) around every rounding boundary of env.py:373-378, as float32-exact a = amps / 32."""
    v = [k / 4 for k in range(0, 129)]                                       # every quarter amp 0 .. 32
    for k in range(0, 32):                                                   # half-integers +- 1 ulp (np.round ties)
        h = np.float32((k + 0.5) / 32)
        v += [float(np.nextafter(h, np.float32(0))) * 32, float(np.nextafter(h, np.float32(1))) * 32]
    six = np.float32(6 / 32)
    v += [float(np.nextafter(six, np.float32(0))) * 32, float(np.nextafter(six, np.float32(1))) * 32]
    for c in (4, 12, 20, 28):                                                # CC: amps / 8 half-integers +- 1 ulp
        h = np.float32(c / 32)
        v += [float(np.nextafter(h, np.float32(0))) * 32, float(np.nextafter(h, np.float32(1))) * 32]
    a = (np.array(v, dtype=np.float64) / 32).astype(np.float32)
    assert len(a) <= 288
    return a


def make_reference_env(envmod, net):
    gen = _Bag(site=net.site, requested_energy_cap=100)
    env = envmod.EVChargingEnv(gen, moer_forecast_steps=36, project_action_in_env=False)   # the real __init__
    env.cn = _Bag(station_ids=list(net.station_ids), min_pilot_signals=np.array(net.min_pilot_signals))
    assert env.num_stations == net.num_stations
    return env


def run_case(envmod, wrmod, net, table, moer, kind, seed, out, name):
    """One episode: the oracle supplies what acnportal would, the reference's methods compute the rest."""
    n, m = net.num_stations, len(net.magnitudes)
    ids = list(net.station_ids)
    arr, dep, est, st, req = table
    sessions = ob.pack_sessions(arr, dep, est, st)
    project = kind == 'project'
    onet = ob.OracleNetwork(net)
    oenv = ob.OracleEnv(onet, 36, project)
    o_obs0 = oenv.reset(sessions, req, moer)

    env = make_reference_env(envmod, net)
    env.moer = moer
    env.t = 0
    for k in env._reward_breakdown:
        env._reward_breakdown[k] = 0.0
    state = {'records': [], 'currents': None, 'pilots': None}
    rates_hist = np.zeros((n, 289))
    env._interface = _Bag(active_sessions=lambda: state['records'])

    def constraint_current(schedule_arr):
        assert np.array_equal(schedule_arr, state['pilots'])      # the stub looks a value up; it computes nothing
        return state['currents']
    env._simulator = _Bag(charging_rates=rates_hist,
                          network=_Bag(magnitudes=np.array(net.magnitudes), constraint_current=constraint_current))
    phase_factor = np.exp(1j * np.deg2rad(net._phase_angles))            # env.py:485
    A_tilde = net.constraint_matrix * phase_factor[None, :]               # env.py:486

    def obs_arrays(o):
        return {k: np.array(o[k], copy=True) for k in ('timestep', 'est_departures', 'demands', 'prev_moer', 'forecasted_moer')}

    obs0 = obs_arrays(env._get_observation())                             # reset(): env.py:338
    rng = np.random.default_rng(seed)
    wrapper = None
    T = 288
    if kind == 'discrete':
        wrapper = object.__new__(wrmod.DiscreteActionWrapper)             # no gymnasium to run __init__'s space checks
        wrapper._bins, wrapper._cont_dtype = 5, env.action_space.dtype
        actions = rng.integers(0, 5, (T, n), dtype=np.int64)
    elif kind == 'boundary':
        vals = boundary_amps()
        T = len(vals)
        actions = np.stack([np.roll(vals, -s)[(np.arange(n) * 5) % len(vals)] for s in range(T)]).astype(np.float32)
    else:
        actions = (rng.random((T, n), dtype=np.float32) ** np.float32(0.5)).astype(np.float32)
        actions[rng.random((T, n)) < 0.05] = 1.0
        actions[rng.random((T, n)) < 0.05] = 0.0
    rec = {k: [] for k in ('a_cont', 'sched_in', 'pilots', 'in_rates', 'in_active', 'in_est', 'in_rem', 'in_currents',
                           'reward', 'breakdown', 'terminated_oracle', 'timestep', 'est_departures', 'demands', 'prev_moer',
                           'forecasted_moer')}
    for t in range(1, T + 1):
        a = actions[t - 1]
        env.t += 1                                                        # env.py:279
        if wrapper is not None:
            a_cont = wrapper.action(a)                                    # wrappers.py:43-45 (reference-run)
            _, res = oenv.step_discrete(a, 5)
        else:
            a_cont = a.copy()
            _, res = oenv.step(a)
        sched_in = np.array(res.projected[:n]) if project else a_cont.copy()   # float64 / float32
        rec['a_cont'].append(np.array(a_cont, copy=True))
        rec['sched_in'].append(sched_in.astype(np.float64))
        sched = env._to_schedule(sched_in.copy())                         # env.py:340-379 (scales its argument in place)
        assert list(sched.keys()) == ids
        pilots = np.array([float(x[0]) for x in sched.values()])
        # --- inputs the reference takes from acnportal: supplied from the oracle's state after the same step ---
        rem, odep, oest = oenv.station_state()
        active = (odep >= 0) & (rem > FULLY_CHARGED_EPS)
        state['records'] = [_Bag(station_id=ids[i], estimated_departure=int(oest[i]), remaining_demand=float(rem[i]))
                            for i in np.flatnonzero(active)]
        rates = np.array(res.rates[:n])
        rates_hist[:, t - 1] = rates
        state['pilots'] = pilots
        state['currents'] = A_tilde @ pilots                              # acnportal constraint_current [MEM], env.py:485-486
        obs = obs_arrays(env._get_observation())                          # env.py:381-394 (reference-run)
        reward = env._get_reward(sched)                                   # env.py:431-464 (reference-run)
        rec['pilots'].append(pilots)
        rec['in_rates'].append(rates)
        rec['in_active'].append(active)
        rec['in_est'].append(np.where(active, oest, 0).astype(np.int16))
        rec['in_rem'].append(np.where(active, rem, 0.0))
        rec['in_currents'].append(state['currents'])
        rec['reward'].append(float(reward))
        rec['breakdown'].append([env._reward_breakdown[k] for k in ('profit', 'carbon_cost', 'excess_charge')])
        rec['terminated_oracle'].append(bool(res.terminated))
        for k, v in obs.items():
            rec[k].append(v)
    pil = np.array(rec['pilots'])
    assert np.array_equal(pil, np.round(pil)) and pil.min() >= 0 and pil.max() <= 32
    key = lambda k: f'{name}|{k}'
    out[key('table_arrival')], out[key('table_departure')], out[key('table_est_departure')] = arr, dep, est
    out[key('table_station')], out[key('table_requested')] = st, req
    out[key('actions')] = actions
    if wrapper is not None:
        out[key('a_cont')] = np.array(rec['a_cont'], dtype=np.float32)      # DiscreteActionWrapper.action's outputs
    if project:
        out[key('sched_in')] = np.array(rec['sched_in'])
    out[key('pilots')] = pil.astype(np.int8)
    out[key('in_rates')] = np.array(rec['in_rates'])
    out[key('in_active')] = np.array(rec['in_active'])
    out[key('in_est')] = np.array(rec['in_est'])
    out[key('in_rem')] = np.array(rec['in_rem'])
    out[key('in_currents')] = np.array(rec['in_currents'])
    out[key('reward')] = np.array(rec['reward'])
    out[key('breakdown')] = np.array(rec['breakdown'])
    out[key('terminated_oracle')] = np.array(rec['terminated_oracle'])
    for k in ('timestep', 'est_departures', 'demands', 'prev_moer', 'forecasted_moer'):
        out[key('obs_' + k)] = np.array(rec[k], dtype=np.float32)
        out[key('obs0_' + k)] = obs0[k].astype(np.float32)
    assert np.array_equal(o_obs0[:n], obs0['demands'])
    return env, rec


def multiagent_case(mamod, envmod, net, rec, out, name):
    """multiagent_env.py:102-148 on a stretch of the case's observations (fresh dicts per step: what a caller that
    copies would pass; the reference itself passes reused buffers, which makes no difference to the result below)."""
    ids = list(net.station_ids)
    env = make_reference_env(envmod, net)
    steps = list(range(100, 106))
    for delay in (0, 3):
        ma = object.__new__(mamod.MultiAgentEVChargingEnv)                 # __init__ builds a whole EVChargingEnv: not needed
        ma.periods_delay, ma.agents, ma.single_env = delay, ids[:], env
        from collections import deque
        ma._past_obs_agg = deque(maxlen=delay)
        got = []
        for j, t in enumerate(steps):
            obs_agg = {k: np.array(rec[k][t], copy=True) for k in ('timestep', 'est_departures', 'demands', 'prev_moer',
                                                                   'forecasted_moer')}
            d = ma._create_dict_from_obs_agg(obs_agg, init=(j == 0))
            assert list(d.keys()) == ids
            got.append(np.stack([d[a] for a in ids]))
        out[f'{name}|ma_delay{delay}'] = np.array(got, dtype=np.float32)
    out[f'{name}|ma_steps'] = np.array(steps)


def main():
    envmod, wrmod, mamod = reference_step_modules()
    eg, ut, lm = reference_generators()
    E = envmod.EVChargingEnv
    out = {}
    names = ('TIMESTEP_DURATION', 'ACTION_SCALE_FACTOR', 'VOLTAGE', 'MARGINAL_REVENUE_PER_KWH', 'OPERATING_MARGIN',
             'MARGINAL_PROFIT_PER_KWH', 'CO2_COST_PER_METRIC_TON', 'A_MINS_TO_KWH', 'VIOLATION_WEIGHT', 'A_PERS_TO_KWH',
             'PROFIT_FACTOR', 'VIOLATION_FACTOR', 'CARBON_COST_FACTOR')
    out['constant_names'] = np.array(names)
    out['constants'] = np.array([float(getattr(E, k)) for k in names])            # env.py:99-114
    # DiscreteActionWrapper.action for every bins 2..9 and every level (wrappers.py:43-45)
    for bins in range(2, 10):
        w = object.__new__(wrmod.DiscreteActionWrapper)
        w._bins, w._cont_dtype = bins, np.dtype(np.float32)
        got = w.action(np.arange(bins, dtype=np.int64))
        assert got.dtype == np.float32
        out[f'discrete_levels_{bins}'] = got
    cases = []
    period = list(ut.DEFAULT_DATE_RANGES)[0]                                        # Summer 2019: the busiest packaged period
    for site in ('caltech', 'jpl'):
        net = site_str_to_site(site)
        import warnings
        warnings.simplefilter('ignore')
        out[f'{site}|station_ids'] = np.array(net.station_ids)
        out[f'{site}|min_pilot_signals'] = np.array(net.min_pilot_signals)
        g = eg.RealTraceGenerator(site, period, sequential=True)
        best, best_n = None, -1
        for seed in range(40):                                                      # the busiest of the first 40 days
            g.set_seed(seed)
            df = g._create_events()
            if len(df) > best_n:
                best, best_n, best_seed = df, len(df), seed
        g.set_seed(best_seed)
        g.get_event_queue()
        moer_real = g.get_moer()
        gm = eg.GMMsTraceGenerator(site, period, seed=11)
        dfs = [gm._create_events() for _ in range(2)]
        moer = np.asarray(moer_real, dtype=np.float64)
        out[f'{site}|moer'] = moer                     # one MOER day per site: the cases share it (and the MOER outputs)
        shared = {}
        for kind, df, seed in (('continuous', best, 1), ('discrete', dfs[0], 2), ('project', dfs[1], 3), ('boundary', best, 4)):
            name = f'{site}|{kind}'
            table = table_of(df, g.station_ids)
            env, rec = run_case(envmod, wrmod, net, table, moer, kind, seed, out, name)
            cases.append(name)
            for k in ('timestep', 'prev_moer', 'forecasted_moer'):      # functions of (t, MOER day) only: stored once per site
                v = out.pop(f'{name}|obs_' + k)
                v0 = out.pop(f'{name}|obs0_' + k)
                if k in shared:
                    assert np.array_equal(v, shared[k][0][:len(v)]) and np.array_equal(v0, shared[k][1])
                else:
                    shared[k] = (v, v0)
                    out[f'{site}|obs_' + k], out[f'{site}|obs0_' + k] = v, v0
            print(name, 'sessions', len(df), 'steps', len(rec['reward']), 'return', sum(rec['reward']))
            if kind == 'continuous':
                multiagent_case(mamod, envmod, net, rec, out, name)
    out['cases'] = np.array(cases)
    # the observation key order spaces.flatten uses ([MEM] gymnasium: sorted keys) as the stub applied it
    out['flatten_key_order'] = np.array(sorted(('timestep', 'est_departures', 'demands', 'prev_moer', 'forecasted_moer')))
    path = os.path.join(HERE, 'step_units.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
