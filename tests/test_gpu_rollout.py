"""The fused rollout kernel (csrc/evc_rollout.h: T periods of a device-resident policy in one launch, state in
registers) against (a) the same rollout as a loop of evc_step launches and (b) the oracle's episode loop
(BaseAlgorithm.run, algorithms/base.py:63-88, under GreedyAlgorithm / RandomAlgorithm, baselines.py:22-51).
Integers — event state, est_departures, terminated, statuses — bit-exact, remaining demands bit-exact against the
step kernels, floats to 1e-9 against the oracle (north_star asks for 1e-5)."""
import os

import numpy as np
import pytest

from oracle import binding as ob
from sustaingym_amd.event_generation import gmm_device_tables
from sustaingym_amd.hostio import to_host
from sustaingym_amd.network import site_str_to_site

pytestmark = pytest.mark.gpu


def _gmm_engine(site, period, N, bank, seed, project=True, autoreset=False, **kw):
    """Engine whose bank holds `bank` device-generated GMM days (the reference's own episode distribution)."""
    from sustaingym_amd.engine import StepEngine
    net = site_str_to_site(site)
    tabs = gmm_device_tables(site, period)
    eng = StepEngine(net, N, project_action=project, autoreset=autoreset, bank_slots=bank, max_sessions=128,
                     moer_days=tabs['num_days'], **kw)
    eng.upload_gmm(tabs)
    eng.upload_moer(_moer_days(site, period))
    eng.generate_episodes(0, bank, seed, 0)
    return net, eng


def _moer_days(site, period):
    from sustaingym_amd.synthetic import synthetic_moer
    return synthetic_moer(gmm_device_tables(site, period)['num_days'], seed=3)


def _ring(eng, policy, bins):
    """A pre-staged action ring for the replay form of evc_rollout: 5 slices; the float ring holds a NaN, a negative and a
    > 1 entry (clamped and flagged like evc_step does)."""
    import torch
    g = torch.Generator(device='cuda')
    g.manual_seed(5)
    if policy == 'ringd':
        return torch.randint(0, bins, (5, eng.N, eng.n), device='cuda', generator=g, dtype=torch.int64)
    a = torch.rand((5, eng.N, eng.n), device='cuda', generator=g, dtype=torch.float32)
    a[1, 3, 2], a[2, 7, 0], a[3, 11, eng.n - 1] = float('nan'), -0.25, 1.5
    return a


def _run(eng, policy, steps, bins, fused):
    import torch
    os.environ['EVC_ROLLOUT_FUSED'] = '1' if fused else '0'
    try:
        if policy in ('ring', 'ringd'):
            out = eng.rollout(actions=_ring(eng, policy, bins), steps=steps, bins=bins if policy == 'ringd' else 0)
        else:
            out = eng.rollout(policy=policy, steps=steps, bins=bins)
        torch.cuda.synchronize()
    finally:
        os.environ.pop('EVC_ROLLOUT_FUSED', None)
    return {k: to_host(v).copy() for k, v in out.items()}


@pytest.mark.parametrize('site,policy,bins,project', [
    ('caltech', 'random', 0, True), ('caltech', 'greedy', 0, True), ('jpl', 'random', 0, True),
    ('jpl', 'greedy', 0, True), ('caltech', 'random', 5, True), ('caltech', 'random', 0, False),
    ('jpl', 'greedy', 0, False), ('caltech', 'ring', 0, True), ('jpl', 'ring', 0, True), ('caltech', 'ringd', 5, True),
    ('caltech', 'ring', 0, False)])
def test_fused_rollout_equals_the_loop_of_steps(site, policy, bins, project):
    """One launch of T periods == T launches of one period: every piece of simulator state and every output,
    for T = 1, a stretch of the congested morning, a whole day, and across an autoreset boundary."""
    N, bank = 1022, 2048                      # not a multiple of 4: the last quad is ragged
    period = 'Summer 2019' if site == 'caltech' else 'Summer 2021'
    engines = []
    for fused in (True, False):
        net, eng = _gmm_engine(site, period, N, bank, seed=77, project=project, autoreset=True)
        eng.set_autoreset_stride(N)
        eng.set_policy_seed(99, env_id_base=5000)
        eng.reset()
        engines.append(eng)
    n = net.num_stations
    total = 0
    for steps in (1, 95, 60, 132, 40, 300):       # 1 + 95 + 60 + 132 = 288: the fourth call ends ON the boundary
        a = _run(engines[0], policy, steps, bins, True)
        b = _run(engines[1], policy, steps, bins, False)
        total += steps
        tag = f'{site} {policy} after {total} steps'
        sa, sb = engines[0].get_state(), engines[1].get_state()
        assert np.array_equal(sa['scalars'], sb['scalars']), tag
        assert np.array_equal(sa['departure'], sb['departure']), tag
        assert np.array_equal(sa['est_departure'], sb['est_departure']), tag
        assert np.array_equal(sa['remaining_kwh'], sb['remaining_kwh']), tag            # same doubles
        np.testing.assert_allclose(sa['breakdown'], sb['breakdown'], rtol=1e-12, atol=1e-13, err_msg=tag)
        assert np.array_equal(a['terminated'], b['terminated']), tag
        assert np.array_equal(a['obs'], b['obs']), tag
        assert np.array_equal(a['final_obs'], b['final_obs']), tag
        np.testing.assert_allclose(a['reward'], b['reward'], rtol=1e-12, atol=1e-14, err_msg=tag)
        np.testing.assert_allclose(a['breakdown'], b['breakdown'], rtol=1e-12, atol=1e-13, err_msg=tag)
        np.testing.assert_allclose(a['returns'], b['returns'], rtol=1e-12, atol=1e-13, err_msg=tag)
    assert (sa['scalars'][:, 7] >= 2).all()               # two episodes finished everywhere
    assert not (sa['scalars'][:, 6] & 2).any()            # EVC_STATUS_PROJ_NOCONV never
    if policy == 'ring':
        assert (sa['scalars'][[3, 7, 11], 6] & 8).all() and int(((sa['scalars'][:, 6] & 8) != 0).sum()) == 3      # EVC_STATUS_ACTION_CLAMPED
    for eng in engines:
        eng.close()


@pytest.mark.parametrize('site,policy,bins', [('caltech', 'greedy', 0), ('caltech', 'random', 0),
                                              ('jpl', 'greedy', 0), ('jpl', 'random', 0), ('caltech', 'random', 5)])
def test_fused_rollout_16384_gmm_days_against_the_oracle(site, policy, bins):
    """whole GMM days at 16 384 environments under greedy / random — episode returns, reward
    breakdowns, the last step's outputs and the final station state equal the oracle's episode loop."""
    N = 16384
    period = 'Summer 2019' if site == 'caltech' else 'Summer 2021'
    net, eng = _gmm_engine(site, period, N, N, seed=1234)
    n = net.num_stations
    ns, sess, req, day, _ = eng.download_episodes(0, N)
    bat = ob.OracleBatch(ob.OracleNetwork(net), N, 36, True)
    bat.set_bank(ns, sess, req, day, _moer_days(site, period))
    eng.set_policy_seed(4321, env_id_base=70000)
    obs0 = to_host(eng.reset()).copy()
    assert np.array_equal(obs0, bat.reset())
    steps = 288
    g = _run(eng, policy, steps, bins, True)
    o = bat.rollout(policy, obs0, steps=steps, bins=bins, seed=4321, env_id_base=70000)
    assert g['terminated'].all() and o['terminated'].all()
    np.testing.assert_allclose(g['returns'], o['returns'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(g['breakdown'], o['breakdown'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(g['reward'], o['reward'], rtol=1e-9, atol=1e-13)
    assert np.array_equal(g['obs'][:, n:], o['obs'][:, n:])                     # est_departures, MOER, timestep
    np.testing.assert_allclose(g['obs'][:, :n], o['obs'][:, :n], rtol=2e-7, atol=0)
    rem, dep, est = eng.station_state()
    o_rem, o_dep, o_est = bat.station_state()
    assert np.array_equal(dep, o_dep) and np.array_equal(est, o_est)
    np.testing.assert_allclose(rem, o_rem, rtol=1e-9, atol=1e-10)
    sc = eng.env_scalars()
    assert (sc['t'] == 288).all() and (sc['episodes'] == 1).all()
    assert np.array_equal(sc['status'] != 0, o['status'] != 0)
    assert not (sc['status'] & 2).any()
    assert g['returns'].std() > 0
    eng.close()


def test_fused_rollout_mid_episode_and_after_done():
    """A rollout that starts mid-episode from a state the step kernel left, runs past the end of the episode
    without autoreset (steps after termination are ignored and flagged), and a zero-demand corner."""
    from helpers import make_workload, make_pair
    from sustaingym_amd.network import caltech_acn
    from sustaingym_amd.hostio import to_device
    net = caltech_acn()
    N, n = 130, net.num_stations
    wl = make_workload(net, N, seed=5, busy=True)
    eng, bat = make_pair(net, N, wl, True, debug=False)
    obs = to_host(eng.reset()).copy()
    assert np.array_equal(obs, bat.reset())
    rng = np.random.default_rng(3)
    for t in range(100):                                   # 100 ordinary steps first
        a = rng.random((N, n), dtype=np.float32)
        g = eng.step(to_device(a))
        o = bat.step(a, debug=False)
    obs = o['obs']
    assert np.array_equal(to_host(g['obs'])[:, n:], obs[:, n:])
    gg = _run(eng, 'greedy', 250, 0, True)                 # 188 live steps + 62 after termination
    oo = bat.rollout('greedy', obs, steps=250)
    np.testing.assert_allclose(gg['returns'], oo['returns'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(gg['breakdown'], oo['breakdown'], rtol=1e-9, atol=1e-12)
    assert gg['terminated'].all() and (gg['reward'] == 0).all()
    sc = eng.env_scalars()
    assert (sc['status'] & 4).all() and (sc['t'] == 288).all()       # EVC_STATUS_STEP_AFTER_DONE
    assert np.array_equal(gg['obs'][:, n:], oo['obs'][:, n:])
    eng.close()


def test_discrete_ring_on_gmm_days_against_the_oracle():
    """The case that exposed a lane-dependent branch in the in-row water-filling (rounds 1-2: the row sums differed in
    their last bits between the lanes of a row, csrc/evc_quad.h row_allreduce_f64): DiscreteActionWrapper levels — equal
    targets, saturated pods — replayed on GMM days, the STEP kernels and the fused kernel against the oracle."""
    site, bins, N, bank, period = 'caltech', 5, 1022, 2048, 'Summer 2019'
    rets = {}
    for fused in (True, False):
        net, eng = _gmm_engine(site, period, N, bank, seed=77, project=True, autoreset=True)
        eng.set_autoreset_stride(N)
        eng.reset()
        ring = to_host(_ring(eng, 'ringd', bins)).copy()
        rets[fused] = [_run(eng, 'ringd', steps, bins, fused)['returns'] for steps in (96, 192, 340)]
        if fused:
            ns, sess, req, day, _ = eng.download_episodes(0, bank)
        assert not (eng.env_scalars()['status'] & 2).any()
        eng.close()
    bat = ob.OracleBatch(ob.OracleNetwork(net), N, 36, True)
    bat.set_bank(ns, sess, req, day, _moer_days(site, period), autoreset_stride=N)
    bat.reset()
    for ci, steps in enumerate((96, 192, 340)):
        ret = np.zeros(N)
        for i in range(steps):
            ret += bat.step(ring[i % 5], bins=bins, autoreset=True, debug=False)['reward']
        for fused in (True, False):
            np.testing.assert_allclose(rets[fused][ci], ret, rtol=1e-9, atol=1e-12, err_msg=f'chunk {ci} fused={fused}')


@pytest.mark.parametrize('site,fused', [('caltech', True), ('caltech', False), ('jpl', True), ('jpl', False)])
def test_discrete_ring_16384_gmm_days_against_the_oracle(site, fused):
    """A whole GMM day of 16 384 environments under replayed DiscreteActionWrapper levels (ties and saturated pods in
    nearly every period), through the fused kernel and through the loop of step kernels: returns, breakdowns, final
    station state against the oracle stepped with the same ring."""
    bins, N = 5, 16384
    period = 'Summer 2019' if site == 'caltech' else 'Summer 2021'
    net, eng = _gmm_engine(site, period, N, N, seed=4321)
    n = net.num_stations
    ns, sess, req, day, _ = eng.download_episodes(0, N)
    bat = ob.OracleBatch(ob.OracleNetwork(net), N, 36, True)
    bat.set_bank(ns, sess, req, day, _moer_days(site, period))
    assert np.array_equal(to_host(eng.reset()), bat.reset())
    ring = to_host(_ring(eng, 'ringd', bins)).copy()
    g = _run(eng, 'ringd', 288, bins, fused)
    ret = np.zeros(N)
    for i in range(288):
        o = bat.step(ring[i % 5], bins=bins, debug=False)
        ret += o['reward']
    np.testing.assert_allclose(g['returns'], ret, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(g['breakdown'], o['breakdown'], rtol=1e-9, atol=1e-12)
    assert np.array_equal(g['obs'][:, n:], o['obs'][:, n:])
    rem, dep, est = eng.station_state()
    o_rem, o_dep, o_est = bat.station_state()
    assert np.array_equal(dep, o_dep) and np.array_equal(est, o_est)
    np.testing.assert_allclose(rem, o_rem, rtol=1e-9, atol=1e-10)
    assert not (eng.env_scalars()['status'] & 2).any()
    eng.close()



def test_both_register_budgets_of_the_rollout_kernel_give_the_same_day():
    """The projecting rollout kernels are built at two register budgets (2 / 3 wavefronts per SIMD) and the engine keeps the
    one that is faster on the caller's workload (launch_rollout, evc_last_rollout_waves): same arithmetic, so the same day
    played from the same state must come out bit for bit the same whichever build ran — and the engine must try both."""
    import torch
    if os.environ.get('EVC_ROLLOUT_WAVES'):
        pytest.skip('EVC_ROLLOUT_WAVES fixes the setting')
    net, eng = _gmm_engine('caltech', 'Summer 2019', 4096, 512, seed=31)
    eng.set_policy_seed(5)
    eng.reset()
    torch.cuda.synchronize()
    snap = eng.get_state()
    seen, first = set(), None
    for rep in range(9):
        eng.set_state(snap)
        out = eng.rollout(policy='random', steps=288)
        torch.cuda.synchronize()                        # a finished launch is what the engine's timing reads at the next one
        seen.add(eng.last_rollout_waves())
        got = {k: to_host(v).copy() for k, v in out.items()}
        got.update(eng.get_state())
        if first is None:
            first = got
        else:
            for k in first:
                assert np.array_equal(first[k], got[k], equal_nan=True), (rep, k, eng.last_rollout_waves())
    assert seen == {2, 3}, seen
    eng.close()


@pytest.mark.parametrize('site', ['caltech', 'jpl'])
def test_caps_shortcut_on_and_off_leave_the_same_state(site, monkeypatch):
    """the fused kernel's caps-only shortcut skips the second evaluation of the rows.  A congested bank (GMM days
    under the greedy policy: pods bind in most periods) played with the shortcut and without it (EVC_CAPS_SHORTCUT=0: every
    undecided environment goes through quad_exact_rows twice like in the step kernels): every piece of state and every
    output bit for bit, no projection flagged as not converged."""
    N = 2048
    period = 'Summer 2019' if site == 'caltech' else 'Summer 2021'
    res = []
    for shortcut in ('1', '0'):
        monkeypatch.setenv('EVC_CAPS_SHORTCUT', shortcut)
        net, eng = _gmm_engine(site, period, N, N, seed=31, autoreset=True)
        eng.set_autoreset_stride(N)
        eng.reset()
        out = _run(eng, 'greedy', 288, 0, True)
        res.append((out, eng.get_state()))
        eng.close()
    (a, sa), (b, sb) = res
    for k in ('scalars', 'departure', 'est_departure', 'remaining_kwh', 'breakdown'):
        assert np.array_equal(sa[k], sb[k]), (site, k)
    for k in ('obs', 'reward', 'terminated', 'breakdown', 'returns', 'final_obs'):
        assert np.array_equal(a[k], b[k]), (site, k)
    assert not (sa['scalars'][:, 6] & 2).any()
