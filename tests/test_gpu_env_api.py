"""Drop-in API layer on the GPU: the conformance properties the reference's own tests assert
(tests/test_evcharging.py: gymnasium check_env / PettingZoo parallel_api_test: spaces, dtypes,
determinism on seed, 288-step episodes, agents == [] after termination) plus parity of the
single-env API against the oracle on REAL trace days."""
import numpy as np
from sustaingym_amd.hostio import to_device, to_host
import pytest

from oracle import binding as ob
from sustaingym_amd import (DiscreteActionWrapper, EVChargingEnv, EVChargingVectorEnv,
                            GMMsTraceGenerator, MultiAgentEVChargingEnv, RealTraceGenerator, SB3VecEnv)
from sustaingym_amd.envs import obs_slices

pytestmark = pytest.mark.gpu


def in_space(space, obs):
    for key, sp in space.spaces.items():
        a = obs[key]
        assert a.dtype == np.float32 and a.shape == sp.shape, key
        assert np.all(a >= sp.low) and np.all(a <= sp.high), key


def test_single_env_conformance_and_determinism():
    env = EVChargingEnv(RealTraceGenerator('caltech', 'Summer 2021'))
    rng = np.random.default_rng(0)
    traj = []
    for rep in range(2):
        obs, info = env.reset(seed=7)
        in_space(env.observation_space, obs)
        assert set(info) == {'max_profit', 'reward_breakdown'}
        assert obs['timestep'][0] == 0 and not obs['demands'].any()
        rng = np.random.default_rng(1)
        rews = []
        for t in range(288):
            a = env.action_space.sample() if rep < 0 else rng.random(54, dtype=np.float32)
            obs, r, term, trunc, info = env.step(a)
            in_space(env.observation_space, obs)
            assert isinstance(r, float) and trunc is False
            assert term == (t == 287)
            rews.append(r)
        traj.append((np.array(rews), dict(info['reward_breakdown']), info['max_profit']))
        with pytest.raises(RuntimeError):
            env.step(a)
    assert np.array_equal(traj[0][0], traj[1][0]) and traj[0][1] == traj[1][1]
    assert abs(sum(traj[0][0]) - (traj[0][1]['profit'] - traj[0][1]['carbon_cost'] - traj[0][1]['excess_charge'])) < 1e-9
    env.close()


@pytest.mark.parametrize('site,period,seed', [('caltech', 'Spring 2020', 2), ('caltech', 'Summer 2019', 40),
                                               ('jpl', 'Fall 2019', 3)])
@pytest.mark.parametrize('project', [True, False])
def test_single_env_real_trace_matches_oracle(site, period, seed, project):
    """EVChargingEnv on a real ACN-Data day vs the CPU oracle fed with the same event table."""
    gen = RealTraceGenerator(site, period)
    env = EVChargingEnv(gen, project_action_in_env=project)
    obs, info = env.reset(seed=seed)
    table, moer = env._evs, env.moer
    onet = ob.OracleNetwork(env.cn)
    orc = ob.OracleEnv(onet, 36, project)
    o_obs = orc.reset(table.sessions, table.requested, moer)
    n = env.num_stations
    sl = obs_slices(n, 36)
    assert abs(info['max_profit'] - ob.max_profit(table.sessions, table.requested)) < 1e-12
    rng = np.random.default_rng(seed)
    for t in range(288):
        a = rng.random(n, dtype=np.float32) if t % 5 else np.ones(n, np.float32)
        obs, r, term, _, info = env.step(a.copy())
        o_obs, res = orc.step(a)
        assert np.array_equal(env._last['pilots'][0], np.array(res.pilots[:n])), t
        assert np.array_equal(obs['est_departures'], o_obs[sl['est_departures']])
        assert np.array_equal(obs['forecasted_moer'], o_obs[sl['forecasted_moer']])
        assert obs['prev_moer'][0] == o_obs[sl['prev_moer']][0] and obs['timestep'][0] == o_obs[sl['timestep']][0]
        np.testing.assert_allclose(obs['demands'], o_obs[sl['demands']], rtol=2e-7)
        assert abs(r - res.reward) <= 1e-9 * max(1e-3, abs(res.reward))
        assert term == bool(res.terminated)
    for key, v in zip(('profit', 'carbon_cost', 'excess_charge'), res.breakdown):
        assert abs(info['reward_breakdown'][key] - v) <= 1e-9 * max(1.0, abs(v))
    env.close()


def test_discrete_wrapper():
    env = DiscreteActionWrapper(EVChargingEnv(RealTraceGenerator('caltech', 'Summer 2021'),
                                              project_action_in_env=False), bins=5)
    assert env.action_space.nvec.tolist() == [5] * 54
    env.reset(seed=1)
    obs, r, term, trunc, info = env.step(np.full(54, 4, dtype=np.int64))
    assert np.array_equal(env.env._last['pilots'][0], np.full(54, 32.0))
    obs, r, term, trunc, info = env.step(np.arange(54) % 5)
    exp = [(8.0 * (i % 5)) for i in range(54)]
    assert env.env._last['pilots'][0].tolist() == exp       # {0,8,16,24,32} legal for AV and CC
    env.close()


def test_multiagent_parallel_api():
    env = MultiAgentEVChargingEnv(GMMsTraceGenerator('caltech', 'Summer 2019'), discrete_bins=-1)
    single = EVChargingEnv(GMMsTraceGenerator('caltech', 'Summer 2019'))
    obss, infos = env.reset(seed=123)
    s_obs, _ = single.reset(seed=123)
    assert env.agents == env.possible_agents and len(env.agents) == 54
    assert env.observation_space(env.agents[0]).shape == (146,)
    assert env.action_space(env.agents[0]).shape == (1,)
    rng = np.random.default_rng(0)
    for t in range(288):
        acts = {a: rng.random(1, dtype=np.float32) for a in env.agents}
        vec = np.array([acts[a][0] for a in env.agents], dtype=np.float32)
        obss, rews, terms, truncs, infos = env.step(acts)
        s_obs, s_r, s_term, _, _ = single.step(vec)
        first = obss[env.possible_agents[0]]
        assert first.shape == (146,) and first.dtype == np.float32
        assert all(o is first for o in obss.values())                      # same array for every agent
        assert np.array_equal(first, single._flat)                          # flatten(Dict) key order
        assert all(abs(r - s_r / 54) < 1e-15 for r in rews.values())
        assert all(v == (t == 287) for v in terms.values()) and not any(truncs.values())
    assert env.agents == []
    # seed determinism (PettingZoo parallel_seed_test)
    o1, _ = env.reset(seed=5)
    o2, _ = env.reset(seed=5)
    assert np.array_equal(o1[env.agents[0]], o2[env.agents[0]])
    env.close()
    single.close()


def test_multiagent_documented_delay():
    env = MultiAgentEVChargingEnv(GMMsTraceGenerator('caltech', 'Summer 2019'), periods_delay=3,
                                  delay_semantics='documented', project_action_in_env=False)
    env.reset(seed=11)
    hist = []
    for t in range(150):
        obss, *_ = env.step({a: np.ones(1, np.float32) for a in env.agents})
        hist.append(env._base._flat.copy())
        if t >= 3:
            a0, i0 = env.agents[0], 0
            o = obss[a0]
            assert o[i0] == hist[t][i0]                                   # own demand: current
            assert np.array_equal(o[1:54], hist[t - 3][1:54])              # others: 3 periods old
            assert np.array_equal(o[108:], hist[t][108:])                  # moer / timestep: current
    env.close()


def test_vector_env_autoreset_and_single_env_equivalence():
    N = 6
    venv = EVChargingVectorEnv(lambda i: GMMsTraceGenerator('caltech', 'Summer 2019'), num_envs=N)
    obs, info = venv.reset(seed=100)
    singles = [EVChargingEnv(GMMsTraceGenerator('caltech', 'Summer 2019')) for _ in range(N)]
    s_obs = [e.reset(seed=100 + i)[0] for i, e in enumerate(singles)]
    for i in range(N):
        assert np.array_equal(obs['forecasted_moer'][i], s_obs[i]['forecasted_moer'])
        assert abs(info['max_profit'][i] - singles[i]._max_profit) < 1e-12
    rng = np.random.default_rng(0)
    for t in range(288 + 20):
        a = rng.random((N, 54), dtype=np.float32)
        obs, rew, term, trunc, info = venv.step(a)
        assert rew.shape == (N,) and term.shape == (N,) and not trunc.any()
        if t < 288:
            for i, e in enumerate(singles):
                so, sr, st, _, si = e.step(a[i])
                assert sr == rew[i] and st == term[i]
                if t < 287:
                    for key in so:
                        assert np.array_equal(so[key], obs[key][i]), (key, t)
                else:
                    assert info['_final_observation'].all()
                    for key in so:
                        assert np.array_equal(so[key], info['final_observation'][key][i]), key
                    assert obs['timestep'][i, 0] == 0                      # first obs of next episode
                    assert abs(info['final_info']['max_profit'][i] - e._max_profit) < 1e-12
        else:
            assert obs['timestep'][0, 0] == np.float32((t - 287) / 288)
    # SB3 VecEnv protocol
    sb3 = SB3VecEnv(venv)
    sb3.seed(3)
    o = sb3.reset()
    assert o['demands'].shape == (N, 54)
    for t in range(288):
        o, r, d, infos = sb3.step(rng.random((N, 54), dtype=np.float32))
    assert d.all() and 'terminal_observation' in infos[0] and r.dtype == np.float32
    assert infos[0]['terminal_observation']['timestep'][0] == 1.0
    av = venv.agent_observations(np.zeros((N, 146), np.float32))
    assert av.shape == (N, 54, 146) and av.strides[1] == 0
    venv.close()
    for e in singles:
        e.close()


def test_greedy_policy_rollout_against_host_loop_and_oracle():
    """GreedyAlgorithm (baselines.py:22-35) three ways on real trace days: (1) a Python episode loop over
    EVChargingEnv (what BaseAlgorithm.run does, base.py:63-88), (2) the oracle driven by the same policy,
    (3) PolicyRollout: one environment per seed, device-resident policy, whole episodes in one evc_rollout."""
    from sustaingym_amd.rollouts import PolicyRollout
    seeds = [40, 41, 42]
    gen = RealTraceGenerator('caltech', 'Summer 2019')
    env = EVChargingEnv(gen)
    host_returns, host_info = [], []
    for seed in seeds:                                    # (1)
        obs, info = env.reset(seed=seed)
        ret, done = 0.0, False
        while not done:
            obs, r, done, _, info = env.step(np.where(obs['demands'] > 0, 1, 0).astype(np.float32))
            ret += r
        host_returns.append(ret)
        host_info.append({'max_profit': info['max_profit'], **info['reward_breakdown']})
    # (2) oracle with the same policy on the first seed
    g2 = RealTraceGenerator('caltech', 'Summer 2019')
    g2.set_seed(seeds[0])
    table, moer = g2.get_event_table(), g2.get_moer()
    orc = ob.OracleEnv(ob.OracleNetwork(env.cn), 36, True)
    o_obs = orc.reset(table.sessions, table.requested, moer)
    ret = 0.0
    for t in range(288):
        o_obs, r = orc.step(np.where(o_obs[:54] > 0, 1, 0).astype(np.float32))
        ret += r.reward
    assert abs(ret - host_returns[0]) <= 1e-9 * max(1.0, abs(ret))
    assert abs(host_info[0]['profit'] - r.breakdown[0]) <= 1e-9 * max(1.0, r.breakdown[0])
    # (3) all seeds in one call
    res = PolicyRollout(RealTraceGenerator('caltech', 'Summer 2019'), 'greedy').run(seeds)
    assert res['seed'] == seeds
    assert np.allclose(res['return'], host_returns, rtol=1e-9, atol=1e-12)
    for i in range(len(seeds)):
        assert abs(res['max_profit'][i] - host_info[i]['max_profit']) < 1e-12
        for key in ('profit', 'carbon_cost', 'excess_charge'):
            assert abs(res['reward_breakdown'][i][key] - host_info[i][key]) <= 1e-9 * max(1.0, abs(host_info[i][key]))
    frame = PolicyRollout(RealTraceGenerator('caltech', 'Summer 2019'), 'greedy').run_frame(2)
    assert list(frame.columns) == ['seed', 'return', 'max_profit', 'reward_breakdown', 'status'] and len(frame) == 2
    env.close()


@pytest.mark.parametrize('bins', [0, 5])
def test_random_policy_bit_exact_and_rollout(bins):
    """EVC_ACTION_RANDOM (baselines.py:38-51 on a counter-based stream): the action rows the device draws are
    bit-identical to the oracle's C statement of the rule — including after an autoreset (episode counter)
    and with a sharding offset — and a whole rollout under the device policy equals the oracle stepped with
    those actions."""
    import torch
    from sustaingym_amd.engine import StepEngine
    from sustaingym_amd.network import caltech_acn
    from helpers import make_workload
    net = caltech_acn()
    N, n, P = 48, net.num_stations, 96
    wl = make_workload(net, N, bank_slots=P, seed=31)
    eng = StepEngine(net, N, project_action=True, autoreset=True, bank_slots=P,
                     max_sessions=wl['sessions'].shape[1], moer_days=wl['moer'].shape[0])
    eng.upload_moer(wl['moer'])
    eng.upload_episodes(wl['n_sessions'], wl['sessions'], wl['requested'], wl['moer_day'])
    eng.set_autoreset_stride(N)
    eng.set_policy_seed(2024, env_id_base=1000)
    bat = ob.OracleBatch(ob.OracleNetwork(net), N, 36, True)
    bat.set_bank(wl['n_sessions'], wl['sessions'], wl['requested'], wl['moer_day'], wl['moer'], autoreset_stride=N)
    g_obs = eng.reset(host=True).copy()
    assert np.array_equal(g_obs, bat.reset())
    ret_o = np.zeros(N)
    ids = 1000 + np.arange(N)
    for t in range(288 + 20):                          # crosses an autoreset boundary
        episode, tt = divmod(t, 288)
        want = ob.random_actions(2024, ids, episode, tt, n, bins)
        got = to_host(eng.fill_random_actions(bins=bins))
        assert np.array_equal(got, want), f't={t}'
        assert got.min() >= 0.0 and got.max() <= (1.0 if bins else np.float32(1.0 - 2.0 ** -24))
        g = eng.step_policy('random', bins=bins)
        o = bat.step(want, autoreset=True, debug=False)
        ret_o += o['reward']
        assert np.array_equal(g['terminated'], o['terminated'])
        assert np.array_equal(g['obs'][:, n:], o['obs'][:, n:])
        np.testing.assert_allclose(g['reward'], o['reward'], rtol=1e-9, atol=1e-13)
    if bins == 0:
        u = np.concatenate([ob.random_actions(2024, ids, 0, tt, n).ravel() for tt in range(40)])
        assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    eng.close()
    # evc_rollout with the device policy = the same episode returns
    eng = StepEngine(net, N, project_action=True, bank_slots=P, max_sessions=wl['sessions'].shape[1],
                     moer_days=wl['moer'].shape[0])
    eng.upload_moer(wl['moer'])
    eng.upload_episodes(wl['n_sessions'], wl['sessions'], wl['requested'], wl['moer_day'])
    eng.set_policy_seed(2024, env_id_base=1000)
    eng.reset()
    out = eng.rollout(policy='random', steps=288, bins=bins)
    torch.cuda.synchronize()
    bat.reset()
    ret = np.zeros(N)
    for tt in range(288):
        ret += bat.step(ob.random_actions(2024, ids, 0, tt, n, bins), debug=False)['reward']
    np.testing.assert_allclose(to_host(out['returns']), ret, rtol=1e-9, atol=1e-12)
    eng.close()


@pytest.mark.parametrize('delay', [0, 3])
def test_multiagent_vector_env_matches_single_multiagent_env(delay):
    """BASELINE config 5 at test size: batched per-agent observations [N, n, F] from the HIP gather
    kernel == the PettingZoo-style single environment (documented delay semantics)."""
    import torch
    from sustaingym_amd import MultiAgentEVChargingVectorEnv
    N, n = 3, 54
    mk = lambda i: GMMsTraceGenerator('caltech', 'Summer 2021')
    venv = MultiAgentEVChargingVectorEnv(mk, num_envs=N, periods_delay=delay, delay_semantics='documented',
                                         project_action_in_env=False, materialize=True)
    obs, _ = venv.reset(seed=50)
    singles = [MultiAgentEVChargingEnv(mk(i), periods_delay=delay, delay_semantics='documented',
                                       project_action_in_env=False) for i in range(N)]
    s_obs = [e.reset(seed=50 + i)[0] for i, e in enumerate(singles)]
    assert obs.shape == (N, n, 146)
    for i in range(N):
        assert np.array_equal(to_host(obs[i, 0]), s_obs[i][singles[i].possible_agents[0]])
    rng = np.random.default_rng(1)
    for t in range(150):
        a = rng.random((N, n), dtype=np.float32)
        obs, rew, term, trunc, info = venv.step(to_device(a))
        got = to_host(obs)
        for i, e in enumerate(singles):
            so, sr, st, _, _ = e.step({ag: a[i, j:j + 1] for j, ag in enumerate(e.possible_agents)})
            for j in (0, 17, 53):
                assert np.array_equal(got[i, j], so[e.possible_agents[j]]), (t, i, j)
            assert abs(float(rew[i, 0]) - sr[e.possible_agents[0]]) < 1e-15
    venv.close()
    for e in singles:
        e.close()


def test_vector_env_torch_output_stays_on_device():
    """output='torch': actions in, observations / rewards out as device tensors (no host copies)."""
    import torch
    N = 8
    venv = EVChargingVectorEnv(lambda i: GMMsTraceGenerator('jpl', 'Fall 2019'), num_envs=N, output='torch')
    host = EVChargingVectorEnv(lambda i: GMMsTraceGenerator('jpl', 'Fall 2019'), num_envs=N)
    obs, info = venv.reset(seed=9)
    hobs, _ = host.reset(seed=9)
    assert obs['demands'].is_cuda and obs['demands'].shape == (N, 52)
    rng = np.random.default_rng(2)
    for t in range(288):
        a = rng.random((N, 52), dtype=np.float32)
        obs, rew, term, trunc, info = venv.step(to_device(a))
        hobs, hrew, hterm, _, hinfo = host.step(a)
        assert rew.is_cuda and term.dtype == torch.bool
        assert np.array_equal(to_host(rew), hrew) and np.array_equal(to_host(term), hterm)
        # after the autoreset at step 288 the two vector envs play different (unseeded) episodes,
        # like two reference envs would after reset(seed=None); compare the terminal observation
        src, hsrc = (obs, hobs) if t < 287 else (info['final_observation'], hinfo['final_observation'])
        for key in hobs:
            assert np.array_equal(to_host(src[key]), hsrc[key]), (key, t)
    assert bool(term.all())
    venv.close()
    host.close()


@pytest.mark.gpu
def test_vector_env_pipeline2_closed_loop_equals_the_default_form():
    """EVChargingVectorEnv(output='torch', pipeline=2) — steps as two half-batch launches, NOT joined by step() — driven by a
    closed-loop policy (the caller's greedy from obs['demands']) computed per half on that half's stream
    (venv.pipeline_halves()): rewards, observations and terminal observations over an episode boundary equal the default
    one-launch-per-step vector env's, bit for bit.  65 536 environments: the smallest batches the engine splits are 32 768."""
    import torch
    from sustaingym_amd.event_generation import DeviceGMMTraceGenerator
    N = 32768
    envs = [EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2019', seed=4), num_envs=N, output='torch', pipeline=p)
            for p in (1, 2)]
    obs = [e.reset(seed=4)[0] for e in envs]
    n = envs[0].num_stations
    acts = [torch.zeros((N, n), dtype=torch.float32, device='cuda') for _ in envs]
    halves = envs[1].pipeline_halves()
    main = torch.cuda.current_stream()
    ret = [torch.zeros(N, dtype=torch.float64, device='cuda') for _ in envs]
    for t in range(300):
        torch.sign(obs[0]['demands'], out=acts[0])
        o, r, term, _, info = envs[0].step(acts[0])
        ret[0] += r
        for sl, st in halves:
            with torch.cuda.stream(st):
                torch.sign(obs[1]['demands'][sl], out=acts[1][sl])
        o2, r2, term2, _, info2 = envs[1].step(acts[1])
        for sl, st in halves:                                  # per-half consumers stay on the half's stream
            with torch.cuda.stream(st):
                ret[1][sl] += r2[sl]
        if t in (0, 150, 286, 287, 288, 299):
            envs[1].join()
            torch.cuda.synchronize()
            assert torch.equal(r, r2) and torch.equal(term, term2), t
            for key in o:
                assert torch.equal(o[key], o2[key]), (key, t)
            if t == 287:
                for key in o:
                    assert torch.equal(info['final_observation'][key], info2['final_observation'][key]), key
    envs[1].join()
    torch.cuda.synchronize()
    assert torch.equal(ret[0], ret[1])
    assert envs[1]._engine.pipelined_steps() >= 290            # the boundary step's bank refill joins; the steps are split
    for e in envs:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize('site,project', [('caltech', True), ('jpl', True), ('caltech', False)])
def test_vector_env_step_policy_greedy_equals_the_callers_greedy(site, project):
    """EVChargingVectorEnv.step(policy='greedy') — the device-resident GreedyAlgorithm (baselines.py:22-35) applied inside the
    lean streaming kernels, no action tensor — against step(sign(obs['demands'])) of a twin environment: observations, rewards,
    terminal observations and the event state bit for bit over an episode boundary, as one launch per step and (at a size the
    engine splits) as pipelined halves; the numpy path (host buffers) too."""
    import torch
    from sustaingym_amd.event_generation import DeviceGMMTraceGenerator
    for N, pipeline, steps in ((4096, 1, 300), (32768, 2, 300)):
        envs = [EVChargingVectorEnv(DeviceGMMTraceGenerator(site, 'Summer 2019', seed=9), num_envs=N, output='torch',
                                    pipeline=pipeline, project_action_in_env=project) for _ in range(2)]
        obs = [e.reset(seed=9)[0] for e in envs]
        n = envs[0].num_stations
        act = torch.zeros((N, n), dtype=torch.float32, device='cuda')
        for t in range(steps):
            envs[1].join()
            torch.sign(obs[1]['demands'], out=act)
            o, r, term, _, info = envs[0].step(policy='greedy')
            o2, r2, term2, _, info2 = envs[1].step(act)
            obs = [o, o2]
            if t in (0, 1, 100, 150, 286, 287, 288, 299):
                for e in envs:
                    e.join()
                torch.cuda.synchronize()
                assert torch.equal(r, r2) and torch.equal(term, term2), (N, t)
                for key in o:
                    assert torch.equal(o[key], o2[key]), (N, key, t)
                if t == 287:
                    for key in o:
                        assert torch.equal(info['final_observation'][key], info2['final_observation'][key]), key
        for e in envs:
            e.join()
        torch.cuda.synchronize()
        sa, sb = envs[0]._engine.get_state(), envs[1]._engine.get_state()
        for key in ('scalars', 'remaining_kwh', 'departure', 'est_departure', 'breakdown', 'entry_rank'):
            if key == 'scalars':                              # status word: the caller's float actions are never clamped either
                assert np.array_equal(sa[key], sb[key]), key
            else:
                assert np.array_equal(sa[key], sb[key]), key
        if pipeline == 2 and project:                         # (without the projection the rule runs in the debug kernels: one launch)
            assert envs[0]._engine.pipelined_steps() >= 290   # the policy form is split into halves like the float32 form
        for e in envs:
            e.close()
    # numpy path: evc_step_host with the policy kind
    envs = [EVChargingVectorEnv(DeviceGMMTraceGenerator(site, 'Summer 2019', seed=3), num_envs=256, project_action_in_env=project)
            for _ in range(2)]
    obs = [e.reset(seed=3)[0] for e in envs]
    for t in range(120):
        a = np.sign(obs[1]['demands']).astype(np.float32)
        o, r, term, _, _ = envs[0].step(policy='greedy')
        o2, r2, term2, _, _ = envs[1].step(a)
        obs = [o, o2]
        assert np.array_equal(r, r2) and np.array_equal(term, term2), t
        for key in o:
            assert np.array_equal(o[key], o2[key]), (key, t)
    for e in envs:
        e.close()


@pytest.mark.gpu
def test_vector_env_step_policy_random_equals_the_staged_random_actions():
    """EVChargingVectorEnv.step(policy='random') — RandomAlgorithm (baselines.py:38-51) on the engine's counter-based stream —
    against a twin that fetches the very actions the policy would apply (StepEngine.fill_random_actions) and hands them in:
    bit for bit over an episode boundary, continuous and DiscreteActionWrapper levels."""
    import torch
    from sustaingym_amd.event_generation import DeviceGMMTraceGenerator
    for bins in (-1, 5):
        envs = [EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2019', seed=2), num_envs=1024, output='torch',
                                    discrete_bins=bins) for _ in range(2)]
        for e in envs:
            e.reset(seed=2)
            e._engine.set_policy_seed(99)
        for t in range(300):
            a = envs[1]._engine.fill_random_actions(bins=max(bins, 0))            # float32 levels in [0, 1]
            o, r, term, _, _ = envs[0].step(policy='random')
            if bins > 0:
                a = torch.round(a * (bins - 1)).to(torch.int64)                   # what a caller of the wrapper passes
            o2, r2, term2, _, _ = envs[1].step(a)
            if t % 37 == 0 or t in (286, 287, 288):
                torch.cuda.synchronize()
                assert torch.equal(r, r2) and torch.equal(term, term2), (bins, t)
                for key in o:
                    assert torch.equal(o[key], o2[key]), (bins, key, t)
        for e in envs:
            e.close()


@pytest.mark.gpu
def test_vector_env_batched_generator_matches_oracle_over_a_boundary():
    """EVChargingVectorEnv fed by one BatchedGMMTraceGenerator: episodes drawn in bulk (the refill
    on a worker thread), two full episodes stepped; every environment is replayed by the oracle
    from the very episodes the sampler produced."""
    from sustaingym_amd.event_generation import BatchedGMMTraceGenerator
    from sustaingym_amd.envs import EVChargingVectorEnv
    from oracle.binding import OracleEnv, OracleNetwork
    from datetime import timedelta
    N = 96
    bg = BatchedGMMTraceGenerator('caltech', 'Summer 2019', seed=5)
    twin = BatchedGMMTraceGenerator('caltech', 'Summer 2019', seed=5)      # same stream, for the oracle
    venv = EVChargingVectorEnv(bg, num_envs=N, project_action_in_env=False, max_sessions=96)
    obs, info = venv.reset()
    first = twin.sample_episodes(N, 96)
    second = twin.sample_episodes(N, 96)
    assert np.allclose(info['max_profit'], first[4])
    onet = OracleNetwork(venv.cn)
    rng = np.random.default_rng(0)
    acts = rng.random((2 * 288, N, 54)).astype(np.float32)
    g_rew = np.zeros((2 * 288, N))
    for t in range(2 * 288):
        obs, rew, term, trunc, info = venv.step(acts[t])
        g_rew[t] = rew
        assert term.all() == (t % 288 == 287)
    for e in range(0, N, 5):
        for ep, drawn in enumerate((first, second)):
            ns, sess, req, day, mp = drawn
            moer = bg.moer_loader.retrieve(bg.date_range[0] + timedelta(days=int(day[e])))
            o = OracleEnv(onet, 36, project=False)
            o.reset(sess[e, :ns[e]], req[e, :ns[e]], moer)
            for t in range(288):
                _, r = o.step(acts[ep * 288 + t, e])
                assert abs(r.reward - g_rew[ep * 288 + t, e]) <= 1e-5 * max(1.0, abs(r.reward))
    venv.close()


@pytest.mark.gpu
def test_vector_env_device_generation_matches_oracle_generator_and_step():
    """EVChargingVectorEnv whose bank is filled on the GPU: the oracle regenerates episode
    numbers 0..3N-1 from (seed, episode) and replays them; rewards agree over two boundaries."""
    from sustaingym_amd.event_generation import DeviceGMMTraceGenerator
    from sustaingym_amd.envs import EVChargingVectorEnv
    from oracle.binding import OracleEnv, OracleGenerator, OracleNetwork
    from datetime import timedelta
    N = 64
    dg = DeviceGMMTraceGenerator('caltech', 'Summer 2019')
    venv = EVChargingVectorEnv(dg, num_envs=N, project_action_in_env=False)
    obs, info = venv.reset(seed=77)
    ns, sess, req, day, mp = OracleGenerator(dg.tables, 54).episodes(77, 0, 3 * N)
    assert np.allclose(info['max_profit'], mp[:N])
    onet = OracleNetwork(venv.cn)
    rng = np.random.default_rng(0)
    T = 2 * 288 + 10
    acts = rng.random((T, N, 54)).astype(np.float32)
    g_rew = np.zeros((T, N))
    for t in range(T):
        obs, rew, term, trunc, info = venv.step(acts[t])
        g_rew[t] = rew
        if t == 287:
            assert term.all() and np.allclose(info['final_info']['max_profit'], mp[:N])
            assert np.allclose(info['max_profit'], mp[N:2 * N])
            # the max_profit values of the slots refilled at THIS boundary stay on the GPU until those episodes are
            # played — the boundary step does not wait for the generating kernel it has just launched
            assert venv._max_profit_pending == [(0, N)]
        if t == 288:
            assert venv._max_profit_pending == [(0, N)]                       # still: this episode reads slots [N, 2N)
    assert venv._max_profit_pending == [(N, N)]                               # second boundary: [0, N) fetched, [N, 2N) refilled
    assert dg.next_episode == 4 * N
    for e in range(0, N, 7):
        for ep in range(3):                       # env e plays episodes e, N+e, 2N+e
            i = ep * N + e
            moer = dg.moer_loader.retrieve(dg.date_range[0] + timedelta(days=int(day[i])))
            o = OracleEnv(onet, 36, project=False)
            o.reset(sess[i, :ns[i]], req[i, :ns[i]], moer)
            for t in range(288 if ep < 2 else 10):
                _, r = o.step(acts[ep * 288 + t, e])
                assert abs(r.reward - g_rew[ep * 288 + t, e]) <= 1e-5 * max(1.0, abs(r.reward))
    venv.close()


@pytest.mark.gpu
def test_vector_env_real_trace_bank_walks_the_days():
    """RealTraceBank (all days of the period resident, days walked inside the kernel): environment i plays
    days (seed + i), (seed + i) + 1, ... mod D; each of those episodes equals a single EVChargingEnv on a
    sequential RealTraceGenerator reset with that day as seed (the reference's way of selecting a day)."""
    from sustaingym_amd.envs import EVChargingEnv, EVChargingVectorEnv
    from sustaingym_amd.event_generation import RealTraceBank, RealTraceGenerator
    N, D = 5, 7
    period = ('2019-05-28', '2019-06-03')                  # 7 days: the walk wraps around
    venv = EVChargingVectorEnv(RealTraceBank('caltech', period), num_envs=N, project_action_in_env=False)
    single = EVChargingEnv(RealTraceGenerator('caltech', period, sequential=True), project_action_in_env=False)
    obs, info = venv.reset(seed=4)
    rng = np.random.default_rng(0)
    acts = rng.random((3 * 288, N, 54)).astype(np.float32)
    v_obs, v_rew, v_mp = [], [], []
    for t in range(3 * 288):
        v_mp.append(info['max_profit'].copy())
        obs, rew, term, _, info = venv.step(acts[t])
        assert term.all() == (t % 288 == 287)
        v_obs.append({k: v.copy() for k, v in (info['final_observation'] if term.all() else obs).items()})
        v_rew.append(rew.copy())
    for i in (0, 3, 4):
        for ep in range(3):
            day = (4 + i + ep) % D
            o, inf = single.reset(seed=day)
            assert abs(inf['max_profit'] - v_mp[ep * 288][i]) < 1e-9
            for t in range(288):
                o, r, term, _, inf = single.step(acts[ep * 288 + t, i])
                assert r == v_rew[ep * 288 + t][i], (i, ep, t)
                for key in o:
                    assert np.array_equal(np.ravel(o[key]), np.ravel(v_obs[ep * 288 + t][key][i])), (i, ep, t, key)
    venv.close(); single.close()


@pytest.mark.gpu
def test_integration_md_binding_stub_runs():
    """The reference-side ctypes stub printed in INTEGRATION.md (Level 2) is executed as written (only the
    library path is substituted) on a stand-in for the acnportal network / EV objects and must reproduce
    the package's own EVChargingEnv step for step."""
    import os
    import re
    import types
    from sustaingym_amd import _lib
    from sustaingym_amd.envs import EVChargingEnv
    from sustaingym_amd.event_generation import GMMsTraceGenerator
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, 'INTEGRATION.md')).read()
    code = re.search(r"```python\n(# sustaingym/envs/evcharging/_hip\.py.*?)```", md, re.S).group(1)
    code = code.replace("C.CDLL('libevcharge_hip.so')", f"C.CDLL({_lib.LIB_PATH!r})")
    _lib.load()                                            # torch's HIP runtime first (see _lib.load)
    ns = {}
    exec(code, ns)
    gen = GMMsTraceGenerator('caltech', 'Summer 2021')
    env = EVChargingEnv(gen, project_action_in_env=True)
    obs, info = env.reset(seed=3)
    # the same episode for the stub: EV stand-ins carrying the attributes the stub reads
    gen2 = GMMsTraceGenerator('caltech', 'Summer 2021')
    gen2.set_seed(3)
    table = gen2.get_event_table()
    moer = gen2.get_moer()
    cn = env.cn
    evs = [types.SimpleNamespace(arrival=int(s['arrival']), departure=int(s['departure']),
                                 estimated_departure=int(s['est_departure']), station_id=cn.station_ids[int(s['station'])],
                                 requested_energy=float(r)) for s, r in zip(table.sessions, table.requested)]
    backend = ns['HipBackend'](cn, 36, True)
    row = backend.reset(evs, moer)
    flat = np.concatenate([np.ravel(obs[k]) for k in sorted(obs)])
    assert np.array_equal(row, flat)
    rng = np.random.default_rng(0)
    for t in range(288):
        a = rng.random(54).astype(np.float32)
        obs, r, term, trunc, info = env.step(a)
        row, r2, done2, bd = backend.step(a)
        assert np.array_equal(row, np.concatenate([np.ravel(obs[k]) for k in sorted(obs)])), t
        assert r == r2 and term == done2
    assert term
    env.close()


@pytest.mark.gpu
def test_torch_lean_path_info_max_profit_follows_the_episode():
    """the lean torch path caches its info dict; after an episode boundary (and after reset) it must hand out the
    NEW episode's max_profit, not the cached array of the previous one (env.py:403-406, 422-429)."""
    import torch
    from sustaingym_amd.event_generation import DeviceGMMTraceGenerator
    N = 256
    venv = EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2019', seed=9), num_envs=N, output='torch')
    venv.reset(seed=9)
    act = torch.full((N, venv.num_stations), 0.5, dtype=torch.float32, device='cuda')
    first = None
    for t in range(1, 292):
        _, _, _, _, info = venv.step(act)
        expect = venv._max_profit[venv._cur_slot]
        if t == 1:
            first = np.array(info['max_profit'])
        if t != 288:                                              # the boundary step's own info reports the NEXT episode's (gymnasium autoreset)
            assert np.array_equal(np.asarray(info['max_profit']), expect), t
        if t == 289:
            assert not np.array_equal(np.asarray(info['max_profit']), first)      # a different episode's values
            assert 'final_observation' not in info
    venv.reset(seed=123)
    _, _, _, _, info = venv.step(act)
    assert np.array_equal(np.asarray(info['max_profit']), venv._max_profit[venv._cur_slot])
    venv.close()


@pytest.mark.gpu
def test_host_step_direct_mode_equals_the_copy_mode(monkeypatch):
    """evc_step_host: the kernels writing the page-locked host buffers themselves (default) against the staged copies
    (EVC_HOST_DIRECT_MAX_BYTES=0): every output of every step, pageable and discrete actions, across an episode boundary."""
    from sustaingym_amd.engine import StepEngine
    from sustaingym_amd.network import caltech_acn
    from helpers import make_workload
    net = caltech_acn()
    N = 300
    wl = make_workload(net, N, bank_slots=64, seed=23, busy=True, moer_days=3)
    engines = []
    for _ in range(2):
        eng = StepEngine(net, N, project_action=True, autoreset=True, bank_slots=64, max_sessions=wl['sessions'].shape[1],
                         moer_days=wl['moer'].shape[0])
        eng.upload_moer(wl['moer'])
        eng.upload_episodes(wl['n_sessions'], wl['sessions'], wl['requested'], wl['moer_day'])
        eng.set_autoreset_stride(5)
        engines.append(eng)
    rng = np.random.default_rng(4)
    assert np.array_equal(engines[0].reset(host=True), engines[1].reset(host=True))
    for t in range(300):
        if t % 3 == 2:
            a, bins = rng.integers(0, 5, (N, net.num_stations)), 5
        else:
            a, bins = rng.random((N, net.num_stations), dtype=np.float32), 0
        monkeypatch.delenv('EVC_HOST_DIRECT_MAX_BYTES', raising=False)
        d = {k: np.array(v) for k, v in engines[0].step(a, bins=bins).items()}
        monkeypatch.setenv('EVC_HOST_DIRECT_MAX_BYTES', '0')
        c = {k: np.array(v) for k, v in engines[1].step(a, bins=bins).items()}
        for key in ('obs', 'reward', 'terminated', 'breakdown'):
            assert np.array_equal(d[key], c[key]), (t, key)
        if d['terminated'].any():
            assert np.array_equal(d['final_obs'], c['final_obs']), t
    monkeypatch.delenv('EVC_HOST_DIRECT_MAX_BYTES', raising=False)
    for eng in engines:
        eng.close()


@pytest.mark.gpu
def test_sb3_lazy_infos_equal_real_dicts():
    """SB3VecEnv(infos='lazy'): the per-environment info objects read the step's batch arrays on access; item for item they
    equal the real dicts of infos='dicts' — within an episode, at its end (terminal_observation) and after it."""
    from sustaingym_amd.event_generation import DeviceGMMTraceGenerator
    N = 96
    envs = [SB3VecEnv(EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2019', seed=6), num_envs=N), infos=mode)
            for mode in ('lazy', 'dicts')]
    for e in envs:
        e.seed(3)
    obs = [e.reset() for e in envs]
    rng = np.random.default_rng(8)
    for t in range(1, 292):
        a = rng.random((N, 54), dtype=np.float32)
        (o1, r1, d1, i1), (o2, r2, d2, i2) = (e.step(a) for e in envs)
        assert np.array_equal(r1, r2) and np.array_equal(d1, d2)
        if t in (1, 150, 288, 289):
            assert len(i1) == N and isinstance(i1[0], dict)
            for k in (0, 17, N - 1):
                lazy, real = i1[k], i2[k]
                assert sorted(lazy.keys()) == sorted(real.keys()) and ('terminal_observation' in lazy) == (t == 288)
                assert lazy['max_profit'] == real['max_profit'] and lazy['reward_breakdown'] == real['reward_breakdown']
                assert lazy.get('TimeLimit.truncated', True) is False and lazy.get('episode') is None
                if t == 288:
                    for key, val in real['terminal_observation'].items():
                        assert np.array_equal(lazy['terminal_observation'][key], val)
                c = lazy.copy()
                assert type(c) is dict and c['max_profit'] == real['max_profit']
    for e in envs:
        e.close()
