"""Shared helpers of the parity tests (test infrastructure)."""
import numpy as np

from oracle import binding as ob
from sustaingym_amd.synthetic import synthetic_episodes, synthetic_moer


def make_workload(net, num_envs, bank_slots=None, seed=0, moer_days=3, busy=False, stride=64,
                  early_fraction=0.05):
    P = bank_slots or num_envs
    kw = dict(min_sessions=30, max_sessions=60, max_arrival=120, min_duration=40,
              max_duration=160) if busy else {}
    n_sessions, sessions, requested, moer_day = synthetic_episodes(
        P, net.num_stations, seed=seed, stride=stride, moer_days=moer_days,
        early_fraction=early_fraction, **kw)
    moer = synthetic_moer(moer_days, seed=seed + 1)
    return dict(n_sessions=n_sessions, sessions=sessions, requested=requested,
                moer_day=moer_day, moer=moer)


def make_pair(net, num_envs, workload, project, autoreset=False, stride=1, k=36, debug=True):
    """(HIP engine, oracle batch) loaded with the same bank."""
    from sustaingym_amd.engine import StepEngine
    P = len(workload['n_sessions'])
    eng = StepEngine(net, num_envs, moer_forecast_steps=k, project_action=project,
                     autoreset=autoreset, bank_slots=P, max_sessions=workload['sessions'].shape[1],
                     moer_days=workload['moer'].shape[0], debug_outputs=debug)
    eng.upload_moer(workload['moer'])
    eng.upload_episodes(workload['n_sessions'], workload['sessions'], workload['requested'],
                        workload['moer_day'])
    eng.set_autoreset_stride(stride)
    onet = ob.OracleNetwork(net)
    bat = ob.OracleBatch(onet, num_envs, k, project)
    bat.set_bank(workload['n_sessions'], workload['sessions'], workload['requested'],
                 workload['moer_day'], workload['moer'], autoreset_stride=stride)
    return eng, bat


def assert_step_parity(g, o, n, tag='', float_rtol=1e-9, check_debug=True):
    """g: engine outputs (numpy), o: oracle outputs.  Integers bit-exact, floats tight."""
    assert np.array_equal(g['terminated'], o['terminated']), tag
    if check_debug:
        # pilots are integer amps: bit-exact
        assert np.array_equal(g['pilots'], o['pilots']), (tag, np.argwhere(g['pilots'] != o['pilots'])[:5])
        np.testing.assert_allclose(g['rates'], o['rates'], rtol=float_rtol, atol=1e-12, err_msg=tag)
        # solver outputs are snapped to a 2^-16 A grid (= 2^-21 normalised): the two solvers agree to
        # ~1e-10 A, so a value may land on adjacent grid points; everything else is far tighter
        np.testing.assert_allclose(g['projected'], o['projected'], rtol=0, atol=2.0 ** -21 + 1e-9, err_msg=tag)
        # beyond last-bit differences (reciprocal-multiply vs divide of the demand cap) only the rare
        # grid flips may remain
        assert np.mean(np.abs(g['projected'] - o['projected']) > 1e-12) < 1e-3, tag
    # observation: est_departures (integers) bit-exact, float32 demands / moer / timestep exact
    # up to a float32 rounding flip of a 1e-13-different float64
    F = g['obs'].shape[1]
    assert np.array_equal(g['obs'][:, n:2 * n], o['obs'][:, n:2 * n]), tag
    assert np.array_equal(g['obs'][:, 2 * n:], o['obs'][:, 2 * n:]), tag
    np.testing.assert_allclose(g['obs'][:, :n], o['obs'][:, :n], rtol=2e-7, atol=0, err_msg=tag)
    np.testing.assert_allclose(g['reward'], o['reward'], rtol=float_rtol, atol=1e-13, err_msg=tag)
    np.testing.assert_allclose(g['breakdown'], o['breakdown'], rtol=float_rtol, atol=1e-12, err_msg=tag)
    assert F == 2 * n + (F - 2 * n)
