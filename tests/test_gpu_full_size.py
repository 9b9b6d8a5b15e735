"""Full BASELINE sizes on the GPU (configs[1..2]: 65 536 batched environments, Caltech and JPL).

Parity at full size is established three ways: (i) a random SAMPLE of environments replayed on the
oracle with the same episodes and actions, (ii) size-independent invariants over the whole batch: the
reward identity reward = profit - carbon - excess summed over the episode, energy conservation of every
battery, the 288-step episode length, and agreement of the two kernel families
(4-environments-per-wavefront vs 1-environment-per-wavefront), and (iii) bench.py's own workload
replayed in full by the oracle for a day and the episode boundary, every output compared."""
import numpy as np
from sustaingym_amd.hostio import to_device, to_host
import pytest

from helpers import make_workload
from oracle import binding as ob

pytestmark = pytest.mark.gpu

A_PERS_TO_KWH = (1 / 60) * (208 / 1000) * 5
PROFIT_FACTOR = A_PERS_TO_KWH * (0.15 * 0.20)


@pytest.mark.parametrize('site', ['caltech', 'jpl'])
def test_full_size_sampled_parity_and_invariants(site, monkeypatch):
    import torch
    from sustaingym_amd.engine import StepEngine
    from sustaingym_amd.network import site_str_to_site
    net = site_str_to_site(site)
    n, N, P, T = net.num_stations, 65536, 2048, 288
    wl = make_workload(net, N, bank_slots=P, seed=31, moer_days=8)

    def make_engine():
        eng = StepEngine(net, N, project_action=True, autoreset=False, bank_slots=P,
                         max_sessions=wl['sessions'].shape[1], moer_days=8)
        eng.upload_moer(wl['moer'])
        eng.upload_episodes(wl['n_sessions'], wl['sessions'], wl['requested'], wl['moer_day'])
        eng.reset()
        return eng
    eng = make_engine()
    monkeypatch.setenv('EVC_KERNEL', 'wave')
    eng_wave = make_engine()
    monkeypatch.delenv('EVC_KERNEL')

    rng = np.random.default_rng(7)
    sample = np.sort(rng.choice(N, 384, replace=False))
    onet = ob.OracleNetwork(net)
    bat = ob.OracleBatch(onet, len(sample), 36, True)
    bat.set_bank(wl['n_sessions'], wl['sessions'], wl['requested'], wl['moer_day'], wl['moer'])
    bat.reset((sample % P).astype(np.int32))
    sidx = to_device(sample)

    gen = torch.Generator(device='cuda')
    gen.manual_seed(99)
    ret = torch.zeros(N, dtype=torch.float64, device='cuda')
    for t in range(T):
        a = torch.rand((N, n), dtype=torch.float32, device='cuda', generator=gen)
        if t % 7 == 0:
            a = (a > 0.3).float()                       # bursts of full-rate requests
        out = eng.step(a)
        out_w = eng_wave.step(a)
        ret += out['reward']
        if t % 24 == 0 or t == T - 1:
            # the two kernel families agree: integer-valued outputs bitwise, floats to reduction order
            assert torch.equal(out['terminated'], out_w['terminated']), t
            assert torch.equal(out['obs'][:, n:], out_w['obs'][:, n:]), t
            assert torch.allclose(out['obs'][:, :n], out_w['obs'][:, :n], rtol=2e-7, atol=0), t
            assert torch.allclose(out['reward'], out_w['reward'], rtol=1e-11, atol=1e-15), t
            assert torch.allclose(out['breakdown'], out_w['breakdown'], rtol=1e-11, atol=1e-13), t
        o = bat.step(to_host(a[sidx]))
        g_obs = to_host(out['obs'][sidx])
        assert np.array_equal(g_obs[:, n:], o['obs'][:, n:]), t
        np.testing.assert_allclose(g_obs[:, :n], o['obs'][:, :n], rtol=2e-7, atol=0)
        np.testing.assert_allclose(to_host(out['reward'][sidx]), o['reward'], rtol=1e-9, atol=1e-13)
        assert np.array_equal(to_host(out['terminated'][sidx]), o['terminated'])
    # ---- invariants over the whole batch ----
    assert bool(out['terminated'].all())                                  # every episode ends at step 288
    bd = out['breakdown']
    assert torch.allclose(ret, bd[:, 0] - bd[:, 1] - bd[:, 2], rtol=1e-9, atol=1e-9)
    sc = eng.env_scalars()
    assert np.all(sc['t'] == 288)
    bits = {b: int(np.sum((sc['status'] & b) != 0)) for b in (1, 2, 4)}
    assert not any(bits.values()), f'status bits set (1=occupied, 2=projection not converged, 4=step after done): {bits}'
    # energy conservation: profit = PROFIT_FACTOR * sum(rates) and delivered energy <= requested
    delivered_kwh = to_host(bd[:, 0]) / (0.15 * 0.20)
    slots = np.arange(N) % P
    requested = np.array([wl['requested'][s, :wl['n_sessions'][s]].sum() for s in range(P)])[slots]
    assert np.all(delivered_kwh <= requested + 1e-6)
    assert delivered_kwh.sum() > 0.2 * requested.sum()                     # the batch actually charged
    rem, dep, est = eng.station_state()
    assert np.all(dep == -1) or np.all(rem[dep != -1] >= -1e-9)
    eng.close()
    eng_wave.close()


@pytest.mark.parametrize('site', ['caltech', 'jpl'])
def test_bench_workload_every_output_against_the_oracle(site):
    """bench.py's own workload (also with --site jpl) at its full size — 65 536 environments, 8192-episode bank, device
    autoreset, projection on, U[0,1) actions — replayed by the oracle (one OpenMP thread per granted CPU
    steps the batch in ~40 ms): every output of every step over one whole day plus the episode boundary."""
    import torch
    from sustaingym_amd.engine import StepEngine
    from sustaingym_amd.network import site_str_to_site
    from sustaingym_amd.synthetic import synthetic_episodes, synthetic_moer
    net = site_str_to_site(site)
    N, n, P = 65536, net.num_stations, 8192
    ns, sess, req, day = synthetic_episodes(P, n, seed=1000, stride=64, moer_days=32)    # = bench.py, rank 0
    moer = synthetic_moer(32, seed=7)
    eng = StepEngine(net, N, project_action=True, autoreset=True, bank_slots=P, max_sessions=64, moer_days=32)
    eng.upload_moer(moer)
    eng.upload_episodes(ns, sess, req, day)
    eng.set_autoreset_stride(1)
    orc = ob.OracleBatch(ob.OracleNetwork(net), N, 36, True)
    orc.set_bank(ns, sess, req, day, moer, autoreset_stride=1)
    slots = (np.arange(N) % P).astype(np.int32)
    assert np.array_equal(eng.reset(slots=slots, host=True), orc.reset(slots))
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1234)
    ring = [torch.rand((N, n), device='cuda', generator=gen) for _ in range(4)]
    ring_h = [to_host(r) for r in ring]
    for t in range(300):
        g = {k: to_host(v) for k, v in eng.step(ring[t % 4]).items()}
        o = orc.step(ring_h[t % 4], autoreset=True, debug=False)
        assert np.array_equal(g['terminated'], o['terminated']), t
        assert np.array_equal(g['obs'][:, n:], o['obs'][:, n:]), t              # est_departures, MOER, timestep
        np.testing.assert_allclose(g['obs'][:, :n], o['obs'][:, :n], rtol=1e-6, atol=1e-6, err_msg=f't={t}')
        np.testing.assert_allclose(g['reward'], o['reward'], rtol=1e-9, atol=1e-12, err_msg=f't={t}')
        if o['terminated'].any():
            m = o['terminated'].astype(bool)
            np.testing.assert_allclose(g['final_obs'][m], o['final_obs'][m], rtol=1e-6, atol=1e-6)
    met = eng.read_metrics()
    assert met['episodes_finished'] == N and met['envs_with_status'] == 0
    eng.close()


def test_config2_4096_caltech_continuous_whole_episode_against_the_oracle(caltech):
    """BASELINE configs[1] at exactly its size: 4 096 batched Caltech environments, continuous actions,
    projection on, one whole episode; EVERY environment replayed by the oracle, every output compared
    (per-station pilots bit-exact, rates / projected actions / rewards to 1e-9)."""
    from helpers import assert_step_parity, make_pair
    N, n = 4096, caltech.num_stations
    wl = make_workload(caltech, N, bank_slots=1024, seed=77, moer_days=8)
    eng, bat = make_pair(caltech, N, wl, project=True)
    slots = (np.arange(N) % 1024).astype(np.int32)
    assert np.array_equal(eng.reset(slots=slots, host=True), bat.reset(slots))
    rng = np.random.default_rng(5)
    for t in range(288):
        a = rng.random((N, n), dtype=np.float32)
        if t % 9 == 0:
            a = (a > 0.25).astype(np.float32)
        assert_step_parity(eng.step(a), bat.step(a), n, tag=f'config 2, step {t + 1}')
    sc = eng.env_scalars()
    assert np.all(sc['t'] == 288) and not (sc['status'] & 7).any()
    eng.close()


def _expected_agent_obs(cur, delayed, n):
    """multiagent_env.py:102-148 with the documented delay semantics, in numpy: agent a of environment e gets
    the flattened observation in which demands / est_departures of the OTHER agents come from `delayed`,
    its own entries and the MOER / timestep part from `cur`.  [N, F] x 2 -> [N, n, F]."""
    N, F = cur.shape
    out = np.broadcast_to(cur[:, None, :], (N, n, F)).copy()
    if delayed is not None:
        out[:, :, :2 * n] = delayed[:, None, :2 * n]
        idx = np.arange(n)
        out[:, idx, idx] = cur[:, :n]
        out[:, idx, n + idx] = cur[:, n:2 * n]
    return out


@pytest.mark.parametrize('delay', [0, 3])
def test_config5_multiagent_8192x54_against_oracle_observations(delay):
    """BASELINE configs[4] at its size: 8 192 environments x 54 agents, materialised per-agent observations
    [8192, 54, 146] (258 MB per step) from the HIP gather kernel + the delay ring of
    MultiAgentEVChargingVectorEnv, across an autoreset boundary.  Expected tensors are built in numpy
    from the ORACLE's observations of the same episodes (integers / MOER bitwise, float32 demands to the usual
    2e-7) and, bitwise, from the history of the engine's own flat observations."""
    import torch
    from sustaingym_amd import DeviceGMMTraceGenerator, MultiAgentEVChargingVectorEnv
    N, n, F = 8192, 54, 146
    gen = DeviceGMMTraceGenerator('caltech', 'Summer 2021', seed=11)
    venv = MultiAgentEVChargingVectorEnv(gen, num_envs=N, periods_delay=delay, delay_semantics='documented',
                                         project_action_in_env=True, materialize=True)
    obs, _ = venv.reset(seed=11)
    eng = venv.venv._engine
    # the same episodes for the oracle: both halves of the double-buffered bank + the period's MOER days
    ns, sess, req, day, _ = eng.download_episodes(0, 2 * N)
    from datetime import timedelta
    g0 = venv.venv.generators[0]
    moer = np.stack([g0.moer_loader.retrieve(venv.venv._day0 + timedelta(days=d)) for d in range(venv.venv._ndays)])
    orc = ob.OracleBatch(ob.OracleNetwork(venv.venv.cn), N, 36, True)
    orc.set_bank(ns, sess, req, day, moer, autoreset_stride=N)
    o_hist = [orc.reset(np.arange(N, dtype=np.int32))]
    g_hist = [to_host(eng.device_outputs()['obs']).copy()]
    assert np.array_equal(g_hist[0], o_hist[0])
    assert tuple(obs.shape) == (N, n, F)
    assert np.array_equal(to_host(obs), _expected_agent_obs(g_hist[0], None, n))
    tgen = torch.Generator(device='cuda')
    tgen.manual_seed(3)
    check_at = set(range(1, 7)) | set(range(284, 296)) | set(range(40, 280, 47))
    for t in range(1, 296):
        a = torch.rand((N, n), device='cuda', generator=tgen)
        obs, rew, term, trunc, info = venv.step(a)
        o = orc.step(to_host(a), autoreset=True, debug=False)
        if t == 288:                                        # autoreset: the histories restart
            assert o['terminated'].all() and bool(term.all())
            o_hist, g_hist = [], []
        o_hist.append(o['obs'])
        g_hist.append(to_host(eng.device_outputs()['obs']).copy())
        np.testing.assert_allclose(to_host(rew[:, 0]), o['reward'] / n, rtol=1e-9, atol=1e-14)
        if t not in check_at:
            continue
        j = len(o_hist) - 1                                 # steps since the last (auto)reset
        dl = None if (delay == 0 or j == 0) else max(0, j - delay)
        got = to_host(obs)
        want_g = _expected_agent_obs(g_hist[j], None if dl is None else g_hist[dl], n)
        assert np.array_equal(got, want_g), f'step {t}: gather kernel / ring vs numpy on the engine observations'
        want_o = _expected_agent_obs(o_hist[j], None if dl is None else o_hist[dl], n)
        assert np.array_equal(got[:, :, n:], want_o[:, :, n:]), f'step {t}: est_departures / MOER / timestep'
        np.testing.assert_allclose(got[:, :, :n], want_o[:, :, :n], rtol=2e-7, atol=0, err_msg=f'step {t}: demands')
    venv.close()
