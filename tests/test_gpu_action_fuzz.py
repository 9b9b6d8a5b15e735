"""Action-distribution fuzz: the suite was green on U[0,1) actions for two rounds while the
in-row water-filling was wrong on the reference's own DiscreteActionWrapper regime (equal targets, saturated pods).
Here the LEAN step kernels (what bench.py times) and the FUSED rollout kernel replay, on congested GMM days of both sites,
the action families where ties and saturation live — DiscreteActionWrapper levels for bins 2 .. 9, all-ones, all-zero,
exact pod-cap ties (every station of a pod at cap / size amps: the row sits exactly ON its limit) and one ulp either
side, two-level mixtures — and every reward, observation and the final station state is compared with the oracle's."""
import numpy as np
import pytest

from oracle import binding as ob
from sustaingym_amd.hostio import to_device, to_host
from test_gpu_rollout import _gmm_engine, _moer_days

pytestmark = pytest.mark.gpu

N, BANK = 1024, 1024


def _families(net, rng):
    """name -> float32 action batch [N, n] (or int64 levels + bins)."""
    n = net.num_stations
    fam = {}
    for bins in range(2, 10):
        fam[f'bins{bins}'] = (rng.integers(0, bins, (N, n), dtype=np.int64), bins)
    fam['ones'] = (np.ones((N, n), np.float32), 0)
    fam['zeros'] = (np.zeros((N, n), np.float32), 0)
    # exact ties on the class caps: every station of a capped class asks for cap / (number of its stations) amps
    kind_cap = {}
    A = np.abs(net.constraint_matrix)
    for c in range(A.shape[0]):
        cols = np.flatnonzero(A[c] > 0)
        phases = np.unique(net.phase_angles[cols])
        if len(phases) == 1 and len(cols) <= 16:                  # a simple row over one class (a pod breaker)
            kind_cap[c] = (cols, net.magnitudes[c] / A[c, cols[0]])
    for eps_name, eps in (('tie', 0), ('tie_up', 1), ('tie_dn', -1)):
        a = rng.random((N, n), dtype=np.float32)
        for c, (cols, cap) in kind_cap.items():
            v = np.float32(cap / len(cols) / 32.0)
            if eps:
                v = np.nextafter(v, np.float32(2.0 if eps > 0 else 0.0))
            a[:, cols] = v
        fam[eps_name] = (a, 0)
    two = np.where(rng.random((N, n)) < 0.5, np.float32(0.25), np.float32(1.0)).astype(np.float32)
    fam['two_level'] = (two, 0)
    fam['uniform'] = (rng.random((N, n), dtype=np.float32), 0)
    return fam


def _check(g, o, n, tag):
    assert np.array_equal(g['terminated'], o['terminated']), tag
    assert np.array_equal(g['obs'][:, n:], o['obs'][:, n:]), tag
    np.testing.assert_allclose(g['obs'][:, :n], o['obs'][:, :n], rtol=2e-7, atol=0, err_msg=tag)
    np.testing.assert_allclose(g['reward'], o['reward'], rtol=1e-9, atol=1e-13, err_msg=tag)
    np.testing.assert_allclose(g['breakdown'], o['breakdown'], rtol=1e-9, atol=1e-12, err_msg=tag)


@pytest.mark.parametrize('site', ['caltech', 'jpl'])
def test_lean_step_kernels_under_tie_and_saturation_action_families(site):
    period = 'Summer 2019' if site == 'caltech' else 'Summer 2021'
    net, eng = _gmm_engine(site, period, N, BANK, seed=202)
    n = net.num_stations
    ns, sess, req, day, _ = eng.download_episodes(0, BANK)
    bat = ob.OracleBatch(ob.OracleNetwork(net), N, 36, True)
    bat.set_bank(ns, sess, req, day, _moer_days(site, period))
    assert np.array_equal(to_host(eng.reset()), bat.reset())
    rng = np.random.default_rng(12)
    fam = _families(net, rng)
    names = list(fam)
    u = rng.random((N, n), dtype=np.float32)
    for t in range(100):                                       # to the congested part of the day
        g = eng.step(to_device(u))
        o = bat.step(u, debug=False)
    for t in range(100, 230):
        name = names[t % len(names)]
        a, bins = fam[name]
        g = {k: to_host(v) for k, v in eng.step(to_device(a), bins=bins).items()}
        o = bat.step(a, bins=bins, debug=False)
        _check(g, o, n, f'{site} step {t + 1} family {name}')
    rem, dep, est = eng.station_state()
    o_rem, o_dep, o_est = bat.station_state()
    assert np.array_equal(dep, o_dep) and np.array_equal(est, o_est)
    np.testing.assert_allclose(rem, o_rem, rtol=1e-9, atol=1e-10)
    assert not (eng.env_scalars()['status'] & 2).any()          # EVC_STATUS_PROJ_NOCONV never
    eng.close()


@pytest.mark.parametrize('site', ['caltech', 'jpl'])
def test_fused_rollout_replays_tie_and_saturation_action_families(site):
    """The same families as a replayed ring through the fused kernel (float families in one ring, each discrete bins value in
    a ring of its own), against the oracle stepped with the same actions."""
    import torch
    period = 'Summer 2019' if site == 'caltech' else 'Summer 2021'
    rng = np.random.default_rng(13)
    net, eng = _gmm_engine(site, period, N, BANK, seed=303)
    n = net.num_stations
    ns, sess, req, day, _ = eng.download_episodes(0, BANK)
    fam = _families(net, rng)
    floats = [v[0] for k, v in fam.items() if v[1] == 0]
    rings = [(np.stack(floats), 0)] + [(np.stack([fam[f'bins{b}'][0], rng.integers(0, b, (N, n), dtype=np.int64)]), b) for b in (2, 3, 5, 9)]
    for ring, bins in rings:
        eng.reset()
        bat = ob.OracleBatch(ob.OracleNetwork(net), N, 36, True)
        bat.set_bank(ns, sess, req, day, _moer_days(site, period))
        bat.reset()
        dev = to_device(ring)
        T = 200
        out = eng.rollout(actions=dev, steps=T, bins=bins)
        torch.cuda.synchronize()
        g = {k: to_host(v).copy() for k, v in out.items()}
        ret = np.zeros(N)
        for t in range(T):
            o = bat.step(ring[t % len(ring)], bins=bins, debug=False)
            ret += o['reward']
        tag = f'{site} ring bins={bins}'
        _check(g, o, n, tag)
        np.testing.assert_allclose(g['returns'], ret, rtol=1e-9, atol=1e-12, err_msg=tag)
        rem, dep, est = eng.station_state()
        o_rem, o_dep, o_est = bat.station_state()
        assert np.array_equal(dep, o_dep) and np.array_equal(est, o_est), tag
        np.testing.assert_allclose(rem, o_rem, rtol=1e-9, atol=1e-10, err_msg=tag)
        assert not (eng.env_scalars()['status'] & 2).any(), tag
    eng.close()
