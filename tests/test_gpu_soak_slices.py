"""Short, fixed-seed slices of the hand-run soaks under tests/soak/ (oracle_soak.py: device-generated GMM days
with projection, lean kernels; network_fuzz.py: random network descriptors), so that every `-m gpu` run
carries a piece of the strongest parity evidence.  ~20 s together."""
import numpy as np
import pytest

from helpers import assert_step_parity, make_pair, make_workload, random_network
from oracle import binding as ob

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('drain', ['auto', '0', '1'])
@pytest.mark.parametrize('site', ['caltech', 'jpl'])
def test_gmm_days_slice(site, drain, monkeypatch):
    """2 048 device-generated GMM days (Summer 2019 model) per site, projection on, lean streaming kernel +
    slow path (as a kernel of its own, drained inside the streaming kernel, and as the engine decides), one whole episode = 590 k env-steps against the oracle: no integer mismatch at all, every
    demand within 1e-6 and every reward within 1e-9 relative, no EVC_STATUS_PROJ_NOCONV."""
    from sustaingym_amd.engine import StepEngine
    from sustaingym_amd.event_generation import gmm_device_tables
    from sustaingym_amd.network import site_str_to_site
    from sustaingym_amd.synthetic import synthetic_moer
    # who solves what the streaming kernel queues: '1' = every workgroup drains its own list inside the
    # streaming kernel (no slow kernel at all), '0' = the slow kernel, 'auto' = the engine's day-long rule
    if drain != 'auto':
        monkeypatch.setenv('EVC_DRAIN', drain)
    net = site_str_to_site(site)
    n, N = net.num_stations, 2048
    tabs = gmm_device_tables(site, 'Summer 2019')
    moer = synthetic_moer(tabs['num_days'], seed=3)
    eng = StepEngine(net, N, project_action=True, autoreset=False, bank_slots=N, max_sessions=128,
                     moer_days=tabs['num_days'])
    eng.upload_moer(moer, 0)
    eng.upload_gmm(tabs)
    eng.generate_episodes(0, N, 4242, 0)
    ns, sess, req, day, _ = eng.download_episodes(0, N)
    bat = ob.OracleBatch(ob.OracleNetwork(net), N, 36, project=True)
    bat.set_bank(ns, sess, req, day, moer)
    slots = np.arange(N, dtype=np.int32)
    assert np.array_equal(eng.reset(slots=slots, host=True), bat.reset(slots))
    rng = np.random.default_rng(17)
    slow, worst = 0, 0.0
    for t in range(288):
        a = rng.random((N, n)).astype(np.float32)
        if t % 40 == 20:
            a[::5] = 1.0
        g, o = eng.step(a), bat.step(a, debug=False)
        assert np.array_equal(g['terminated'], o['terminated']), t
        assert np.array_equal(g['obs'][:, n:], o['obs'][:, n:]), t
        d = np.abs(g['obs'][:, :n] - o['obs'][:, :n]) / np.maximum(np.abs(o['obs'][:, :n]), 1e-3)
        assert d.max() <= 1e-6, (t, d.max())
        r = np.abs(g['reward'] - o['reward']) / np.maximum(np.abs(o['reward']), 1e-3)
        worst = max(worst, float(r.max()))
        assert worst <= 1e-9, (t, worst)
        slow += eng.last_slow_count()
    assert slow > 0, 'the slow kernel never ran: the slice does not cover the congested regime'
    assert not (eng.env_scalars()['status'] & 2).any()
    assert not (o['status'] & 2).any()
    eng.close()


@pytest.mark.parametrize('case_seed', [3, 11])
def test_random_network_slice(case_seed):
    """Two fixed random network descriptors (tests/soak/network_fuzz.py's generator) through the debug and the
    lean kernels against the oracle for a whole day, projection on: every output, and neither side may flag
    a projection that did not converge."""
    rng = np.random.default_rng(case_seed)
    net = random_network(rng, f'slice{case_seed}')
    n, N = net.num_stations, 64
    wl = make_workload(net, N, seed=200 + case_seed, busy=True, stride=96)
    eng, bat = make_pair(net, N, wl, project=True, debug=True)
    lean, _ = make_pair(net, N, wl, project=True, debug=False)
    assert np.array_equal(eng.reset(host=True), bat.reset())
    lean.reset(host=True)
    arng = np.random.default_rng(case_seed)
    for t in range(288):
        a = arng.random((N, n), dtype=np.float32)
        if t % 50 == 25:
            a[::3] = 1.0
        g, o = eng.step(a), bat.step(a)
        assert_step_parity(g, o, n, tag=f'network seed {case_seed} n={n} m={len(net.magnitudes)} t={t}')
        l = lean.step(a)
        assert np.array_equal(l['terminated'], g['terminated'])
        np.testing.assert_allclose(l['obs'], g['obs'], rtol=0, atol=2e-5)
        np.testing.assert_allclose(l['reward'], g['reward'], rtol=1e-11, atol=1e-13)
    assert not (eng.env_scalars()['status'] & 2).any() and not (o['status'] & 2).any()
    eng.close()
    lean.close()
