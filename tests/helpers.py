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


def make_pair(net, num_envs, workload, project, autoreset=False, stride=1, k=36, debug=True,
              charge_calculation='continuous'):
    """(HIP engine, oracle batch) loaded with the same bank."""
    from sustaingym_amd.engine import StepEngine
    P = len(workload['n_sessions'])
    eng = StepEngine(net, num_envs, moer_forecast_steps=k, project_action=project,
                     autoreset=autoreset, bank_slots=P, max_sessions=workload['sessions'].shape[1],
                     moer_days=workload['moer'].shape[0], debug_outputs=debug,
                     charge_calculation=charge_calculation)
    eng.upload_moer(workload['moer'])
    eng.upload_episodes(workload['n_sessions'], workload['sessions'], workload['requested'],
                        workload['moer_day'])
    eng.set_autoreset_stride(stride)
    onet = ob.OracleNetwork(net)
    bat = ob.OracleBatch(onet, num_envs, k, project, charge_calculation)
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


def random_network(rng, tag='fuzz'):
    """A random network descriptor: station count 3..64 (odd ones too), 1..14 station classes (every WORDS
    instantiation), 1..16 rows — class caps ("pods") first, then rows over several classes with mixed
    signs —, mixed AV / CC EVSEs.  Used by tests/soak/network_fuzz.py and test_gpu_synthetic_network.py."""
    from sustaingym_amd.network import ChargingNetwork
    n = int(rng.integers(3, 65)); G = int(rng.integers(1, min(14, n) + 1)); m = int(rng.integers(1, 17))
    cls = np.sort(rng.integers(0, G, n)); cls[:G] = np.arange(G); cls = np.sort(cls)
    G = len(np.unique(cls))
    phase_of = rng.choice([30.0, -90.0, 150.0], size=G)
    rows, mags = [], []
    for r in range(m):
        coef = np.zeros(G)
        if r < min(G, m // 2 + 1):                     # class caps ("pods"): simple rows
            coef[r] = 1.0
            size = int((cls == r).sum())
            mags.append(max(20.0, 32.0 * size * float(rng.uniform(0.3, 0.9))))
        else:
            pick = rng.choice(G, size=min(G, int(rng.integers(1, 5))), replace=False)
            coef[pick] = rng.choice([1.0, -1.0, 0.5, 0.25], size=len(pick))
            load = float(np.abs(coef[cls]).sum()) * 32.0
            mags.append(max(20.0, load * float(rng.uniform(0.25, 0.8))))
        rows.append(coef[cls])
    return ChargingNetwork(site=tag, station_ids=[f'S{i:02d}' for i in range(n)],
                           constraint_matrix=np.array(rows), phase_angles=phase_of[cls], magnitudes=np.array(mags),
                           constraint_names=[f'r{i}' for i in range(m)],
                           evse_kind=(rng.random(n) < 0.2).astype(np.uint8))
