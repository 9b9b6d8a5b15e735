"""The Python half of the reference's step, REFERENCE-RUN: tests/golden/step_units.npz holds the outputs of the
reference's own ``EVChargingEnv._to_schedule`` / ``_get_observation`` / ``_get_reward`` / class constants,
``DiscreteActionWrapper.action`` and ``MultiAgentEVChargingEnv._create_dict_from_obs_agg`` (run unmodified from
/root/reference by tests/golden/make_step_unit_golden.py, which says what is stubbed — containers only — and which
values are inputs taken from the oracle because acnportal / cvxpy are not in the image).  Here the same episodes are
replayed through the oracle (CPU) and the HIP engine (-m gpu, through the C-ABI) and every reference-run output is
compared: pilots / est_departures / MOER / timestep bit-exact, float32 demands to one ulp, rewards to 1e-9 relative
(north_star asks for 1e-5).  Pins SURVEY §8a rows a1, a2 and the reference-side halves of a8 / a9 / a14."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
Z = np.load(os.path.join(HERE, 'golden', 'step_units.npz'), allow_pickle=False)
CASES = [str(c) for c in Z['cases']]
REWARD_RTOL = 1e-9


def _case(name):
    site, kind = name.split('|')
    g = lambda k: Z[f'{name}|{k}']
    s = lambda k: Z[f'{site}|{k}']
    T = len(g('reward'))
    from oracle.binding import pack_sessions
    c = dict(site=site, kind=kind, T=T, project=kind == 'project', bins=5 if kind == 'discrete' else 0,
             sessions=pack_sessions(g('table_arrival'), g('table_departure'), g('table_est_departure'), g('table_station')),
             requested=g('table_requested'), moer=s('moer'), actions=g('actions'), pilots=g('pilots').astype(np.float64),
             in_rates=g('in_rates'), reward=g('reward'), breakdown=g('breakdown'), demands=g('obs_demands'),
             est=g('obs_est_departures'), forecast=s('obs_forecasted_moer')[:T], prev=s('obs_prev_moer')[:T],
             timestep=s('obs_timestep')[:T], in_currents=g('in_currents'))
    c['obs0'] = np.concatenate([g('obs0_demands'), g('obs0_est_departures'), s('obs0_forecasted_moer'), s('obs0_prev_moer'),
                                s('obs0_timestep')])
    return c


def _network(site):
    import warnings
    from sustaingym_amd.network import site_str_to_site
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        net = site_str_to_site(site)
    assert list(net.station_ids) == [str(x) for x in Z[f'{site}|station_ids']]
    assert np.array_equal(net.min_pilot_signals, Z[f'{site}|min_pilot_signals'])
    return net


def _check_step(c, t, n, obs, reward, breakdown, pilots=None, rates=None, tag='', strict_pilots=True):
    """Every reference-run output of step t + 1 against one implementation's."""
    k = c['forecast'].shape[1]
    tag = f'{tag} step {t + 1}'
    assert np.array_equal(obs[n:2 * n], c['est'][t]), tag                                   # env.py:387, integers
    assert np.array_equal(obs[2 * n:2 * n + k], c['forecast'][t]), tag                      # env.py:391
    assert obs[2 * n + k] == c['prev'][t][0] and obs[2 * n + k + 1] == c['timestep'][t][0], tag   # env.py:390, 392
    np.testing.assert_allclose(obs[:n], c['demands'][t], rtol=1.2e-7, atol=0, err_msg=tag)  # env.py:388, float32
    flips = 0.0
    if pilots is not None:
        if strict_pilots:
            assert np.array_equal(pilots, c['pilots'][t]), (tag, np.flatnonzero(pilots != c['pilots'][t]))   # env.py:373-378
        flips = float(np.mean(pilots != c['pilots'][t]))
    if rates is not None:
        np.testing.assert_allclose(rates, c['in_rates'][t], rtol=1e-9, atol=1e-12, err_msg=tag)
    if flips == 0.0:
        np.testing.assert_allclose(reward, c['reward'][t], rtol=REWARD_RTOL, atol=1e-14, err_msg=tag)   # env.py:431-464
    return flips


def test_constants_are_the_reference_class_constants():
    """env.py:99-114 read off the reference class, against the host mirror (whose values reach the kernels' constants
    through the same expressions) — bit for bit."""
    from sustaingym_amd.envs import EVChargingEnv
    for name, value in zip(Z['constant_names'], Z['constants']):
        assert float(getattr(EVChargingEnv, str(name))) == float(value), name


def test_discrete_wrapper_levels_are_the_reference_ones():
    """wrappers.py:43-45 for bins 2 .. 9, every level: the host wrapper and the oracle's float32 division."""
    from sustaingym_amd.envs import DiscreteActionWrapper
    for bins in range(2, 10):
        ref = Z[f'discrete_levels_{bins}']
        w = object.__new__(DiscreteActionWrapper)
        w._bins = bins
        got = w.action(np.arange(bins, dtype=np.int64))
        assert got.dtype == np.float32 and np.array_equal(got, ref), bins
        assert np.array_equal((np.arange(bins, dtype=np.float32) / np.float32(bins - 1)), ref)


@pytest.mark.parametrize('name', CASES)
def test_oracle_against_the_reference_run_step_units(name):
    from oracle import binding as ob
    c = _case(name)
    net = _network(c['site'])
    n = net.num_stations
    env = ob.OracleEnv(ob.OracleNetwork(net), 36, c['project'])
    obs = env.reset(c['sessions'], c['requested'], c['moer'])
    assert np.array_equal(obs, c['obs0']), name
    A_tilde = net.constraint_matrix * np.exp(1j * np.deg2rad(net._phase_angles))[None, :]
    for t in range(c['T']):
        a = c['actions'][t]
        obs, r = env.step_discrete(a, 5) if c['bins'] else env.step(a)
        pilots = np.array(r.pilots[:n])
        _check_step(c, t, n, obs, r.reward, None, pilots, np.array(r.rates[:n]), tag=name)
        np.testing.assert_allclose(np.array(r.breakdown), c['breakdown'][t], rtol=REWARD_RTOL, atol=1e-13)
        # the constraint currents the fixture handed to _get_reward are what its own schedule implies (the input is sound)
        np.testing.assert_allclose(A_tilde @ pilots, c['in_currents'][t], rtol=1e-13, atol=1e-12)
    if c['kind'] == 'discrete':
        # the wrapper's float32 levels the reference produced are what the oracle's discrete entry point divides out
        assert np.array_equal(Z[f'{name}|a_cont'], c['actions'].astype(np.float32) / np.float32(4))


def test_projected_values_are_scaled_and_rounded_like_the_reference():
    """'project' cases: ``_to_schedule`` of the reference received the ORACLE's projected action (float64, moved values on
    the tie-snap grid) — its pilots equal the oracle's on every station, i.e. env.py:366-378 on projected values."""
    for name in CASES:
        if not name.endswith('project'):
            continue
        x = Z[f'{name}|sched_in']
        pil = Z[f'{name}|pilots']
        moved = np.abs(x - Z[f'{name}|actions'].astype(np.float64)) > 0
        assert moved.mean() > 0.01, 'the case must exercise the projection'
        kind = _network(name.split('|')[0]).evse_kind
        amps = x * 32
        want = np.where(kind[None, :] == 0, np.where(amps >= 6, np.round(amps), 0), np.round(amps / 8) * 8)
        assert np.array_equal(want, pil)


def test_multiagent_dict_is_the_current_observation_for_every_agent():
    """multiagent_env.py:102-148, reference-run with periods_delay 0 AND 3: every agent's row equals the current flattened
    observation (with a delay the reference writes every agent's own current value into ONE shared array, so nothing
    is delayed — SURVEY Appendix B).  That is what ``delay_semantics='reference'`` (the default) hands out."""
    for name in CASES:
        if not name.endswith('continuous'):
            continue
        c = _case(name)
        steps = Z[f'{name}|ma_steps']
        for delay in (0, 3):
            got = Z[f'{name}|ma_delay{delay}']                       # [steps, agents, F]
            for j, t in enumerate(steps):
                flat = np.concatenate([c['demands'][t], c['est'][t], c['forecast'][t], c['prev'][t], c['timestep'][t]])
                assert np.array_equal(got[j], np.broadcast_to(flat, got[j].shape)), (name, delay, t)
    assert [str(k) for k in Z['flatten_key_order']] == ['demands', 'est_departures', 'forecasted_moer', 'prev_moer', 'timestep']


@pytest.mark.gpu
@pytest.mark.parametrize('debug', [True, False])
@pytest.mark.parametrize('name', CASES)
def test_hip_engine_against_the_reference_run_step_units(name, debug):
    """The same replay through the C-ABI: debug kernels (per-station pilots and delivered amps) and the lean production
    kernels (observation, reward, breakdown only)."""
    from sustaingym_amd.engine import StepEngine
    c = _case(name)
    net = _network(c['site'])
    n = net.num_stations
    eng = StepEngine(net, 1, project_action=c['project'], bank_slots=1, max_sessions=max(1, len(c['sessions'])),
                     moer_days=1, debug_outputs=debug)
    eng.upload_moer(c['moer'][None])
    eng.upload_episodes(np.array([len(c['sessions'])], np.int32), c['sessions'][None], c['requested'][None],
                        np.zeros(1, np.int32))
    obs = eng.reset(host=True)
    assert np.array_equal(obs[0], c['obs0']), name
    flips = []
    for t in range(c['T']):
        g = eng.step(np.ascontiguousarray(c['actions'][t][None]), bins=c['bins'])
        f = _check_step(c, t, n, g['obs'][0], g['reward'][0], None, g['pilots'][0] if debug else None,
                        g['rates'][0] if debug else None, tag=f'{name} debug={debug}', strict_pilots=not c['project'])
        flips.append(f)
        if f == 0.0:
            np.testing.assert_allclose(g['breakdown'][0], c['breakdown'][t], rtol=1e-7 if c['project'] else REWARD_RTOL, atol=1e-12)
    # with the projection the two solvers may land on adjacent grid points of the tie snap: < 1e-3 of pilots
    assert np.mean(flips) < 1e-3
    eng.close()


@pytest.mark.gpu
def test_multiagent_vector_env_rows_against_the_reference_run_dict(caltech):
    """The batched per-agent observation (zero-copy view, reference semantics) holds, for every agent, the row the reference's
    ``_create_dict_from_obs_agg`` produced — here: equal to the engine's current flat observation, which the test above
    pins against the reference's ``_get_observation``."""
    from sustaingym_amd.engine import StepEngine
    name = 'caltech|continuous'
    c = _case(name)
    n = caltech.num_stations
    eng = StepEngine(caltech, 1, project_action=False, bank_slots=1, max_sessions=len(c['sessions']), moer_days=1)
    eng.upload_moer(c['moer'][None])
    eng.upload_episodes(np.array([len(c['sessions'])], np.int32), c['sessions'][None], c['requested'][None], np.zeros(1, np.int32))
    eng.reset(host=True)
    steps = [int(t) for t in Z[f'{name}|ma_steps']]
    for t in range(max(steps) + 1):
        g = eng.step(np.ascontiguousarray(c['actions'][t][None]))
        if t in steps:
            j = steps.index(t)
            for delay in (0, 3):
                ref = Z[f'{name}|ma_delay{delay}'][j]
                assert np.array_equal(ref[:, n:], np.broadcast_to(g['obs'][0][n:], ref[:, n:].shape))
                np.testing.assert_allclose(ref[:, :n], np.broadcast_to(g['obs'][0][:n], ref[:, :n].shape), rtol=1.2e-7)
    eng.close()
