"""The lean streaming kernels exist with the shape of the reference's two sites compiled in (54 / 52 stations, forecast horizon 36:
step_kernel_cquad's NC) and in the general form that takes the shape from Params.  Same bits, and the same as the debug kernels
(general form, the ones the oracle parity tests drive)."""
import numpy as np
import pytest

from helpers import make_workload

pytestmark = pytest.mark.gpu


def _engine(net, N, wl, project, debug, k=36):
    from sustaingym_amd.engine import StepEngine
    eng = StepEngine(net, N, moer_forecast_steps=k, project_action=project, autoreset=True, bank_slots=len(wl['n_sessions']),
                     max_sessions=wl['sessions'].shape[1], moer_days=wl['moer'].shape[0], debug_outputs=debug)
    eng.upload_moer(wl['moer'])
    eng.upload_episodes(wl['n_sessions'], wl['sessions'], wl['requested'], wl['moer_day'])
    eng.set_autoreset_stride(3)
    return eng


@pytest.mark.parametrize('N', [1000, 1001])         # whole quads (the ALIVE copies, with autoreset) / a ragged last quad (the copies that test `env < N`)
@pytest.mark.parametrize('busy', [False, True])
@pytest.mark.parametrize('project', [True, False])
@pytest.mark.parametrize('site', ['caltech', 'jpl'])
def test_site_kernels_equal_the_general_kernels(site, project, busy, N, caltech, jpl, monkeypatch):
    net = caltech if site == 'caltech' else jpl
    n = net.num_stations
    wl = make_workload(net, N, bank_slots=64, seed=21, busy=busy)
    site_eng = _engine(net, N, wl, project, debug=False)
    monkeypatch.setenv('EVC_SITE_KERNELS', '0')     # read when an engine is created
    gen_eng = _engine(net, N, wl, project, debug=False)
    monkeypatch.delenv('EVC_SITE_KERNELS')
    dbg_eng = _engine(net, N, wl, project, debug=True)
    slots = (np.arange(N) * 7) % 64
    obs = [e.reset(slots=slots, host=True).copy() for e in (site_eng, gen_eng, dbg_eng)]
    assert np.array_equal(obs[0], obs[1]) and np.array_equal(obs[0], obs[2])
    rng = np.random.default_rng(5)
    for t in range(300):                            # past the episode boundary: autoreset into the next slots
        a = rng.random((N, n), dtype=np.float32) * (1.3 if t % 7 == 0 else 1.0) - (0.1 if t % 11 == 0 else 0.0)
        outs = [e.step(a) for e in (site_eng, gen_eng, dbg_eng)]
        for key in ('obs', 'reward', 'terminated'):
            assert np.array_equal(outs[0][key], outs[1][key]), (key, t)
            assert np.array_equal(outs[0][key], outs[2][key]), (key, t, 'debug kernel')
    sc = [e.env_scalars() for e in (site_eng, gen_eng, dbg_eng)]
    for key in sc[0]:
        assert np.array_equal(sc[0][key], sc[1][key]) and np.array_equal(sc[0][key], sc[2][key]), key
    for e in (site_eng, gen_eng, dbg_eng):
        e.close()


def test_other_horizon_takes_the_general_kernel(caltech):
    """54 stations but k = 12: Params does not describe the compiled-in shape, the general kernel runs (and agrees with the debug kernel)."""
    N, n = 256, caltech.num_stations
    wl = make_workload(caltech, N, bank_slots=32, seed=4)
    lean, dbg = _engine(caltech, N, wl, True, False, k=12), _engine(caltech, N, wl, True, True, k=12)
    o0, o1 = lean.reset(host=True).copy(), dbg.reset(host=True).copy()
    assert np.array_equal(o0, o1) and o0.shape[1] == 2 * n + 12 + 2
    rng = np.random.default_rng(1)
    for t in range(60):
        a = rng.random((N, n), dtype=np.float32)
        g, d = lean.step(a), dbg.step(a)
        for key in ('obs', 'reward', 'terminated'):
            assert np.array_equal(g[key], d[key]), (key, t)
    lean.close(); dbg.close()


def test_clocks_behind_the_episode_end_leave_the_alive_kernels(caltech):
    """The ALIVE copies assume no environment stands behind its episode's end; autoreset keeps it so — unless somebody writes
    such clocks into the scalars (a restored checkpoint of an engine without autoreset, a hand-made state).  The engine notices
    (evc_set_env_scalars) and runs the copies that test it: rows with t >= 288 report reward 0 / terminated and keep their state,
    exactly as the debug kernel (general form) has it.  A non-autoreset engine stepping past the end: the same."""
    N, n = 512, caltech.num_stations
    wl = make_workload(caltech, N, bank_slots=32, seed=9)
    lean, dbg = _engine(caltech, N, wl, True, False), _engine(caltech, N, wl, True, True)
    for e in (lean, dbg):
        e.reset(host=True)
    rng = np.random.default_rng(2)
    acts = rng.random((40, N, n), dtype=np.float32)
    for t in range(20):
        g, d = lean.step(acts[t]), dbg.step(acts[t])
    st = lean.get_state()
    st['scalars'] = st['scalars'].copy()
    st['scalars'][::3, 0] = 288                      # every third environment: behind the end
    st['scalars'][1::7, 0] = 300
    for e in (lean, dbg):
        e.set_state(st)
    for t in range(20, 40):
        g, d = lean.step(acts[t]), dbg.step(acts[t])
        for key in ('obs', 'reward', 'terminated'):
            assert np.array_equal(g[key], d[key]), (key, t)
        assert (g['reward'][::3] == 0).all() and g['terminated'][::3].all()
    sc0, sc1 = lean.env_scalars(), dbg.env_scalars()
    for key in sc0:
        assert np.array_equal(sc0[key], sc1[key]), key
    lean.close(); dbg.close()
    # without autoreset: past the end of the episode
    from sustaingym_amd.engine import StepEngine
    engs = []
    for debug in (False, True):
        e = StepEngine(caltech, N, moer_forecast_steps=36, project_action=True, autoreset=False, bank_slots=32,
                       max_sessions=wl['sessions'].shape[1], moer_days=wl['moer'].shape[0], debug_outputs=debug)
        e.upload_moer(wl['moer']); e.upload_episodes(wl['n_sessions'], wl['sessions'], wl['requested'], wl['moer_day'])
        e.reset(host=True)
        engs.append(e)
    for t in range(292):
        a = acts[t % 40]
        g, d = engs[0].step(a), engs[1].step(a)
        for key in ('obs', 'reward', 'terminated'):
            assert np.array_equal(g[key], d[key]), (key, t)
    assert g['terminated'].all() and (g['reward'] == 0).all()
    for e in engs:
        e.close()


@pytest.mark.parametrize('policy', ['greedy', 'random', 'ring'])
@pytest.mark.parametrize('site', ['caltech', 'jpl'])
def test_fused_rollout_site_kernels_equal_the_general_kernels(site, policy, caltech, jpl, monkeypatch):
    """evc_rollout's kernels exist with the site's shape and the all-alive predicates compiled in (rollout_kernel's NC / ALIVE) and
    in the general form: same outputs, returns and state over 300 periods (across the episode boundary), on busy days."""
    import torch
    net = caltech if site == 'caltech' else jpl
    N, n = 1024, net.num_stations
    wl = make_workload(net, N, bank_slots=64, seed=33, busy=True)
    fast = _engine(net, N, wl, True, debug=False)
    monkeypatch.setenv('EVC_SITE_KERNELS', '0')
    general = _engine(net, N, wl, True, debug=False)
    monkeypatch.delenv('EVC_SITE_KERNELS')
    outs = []
    for e in (fast, general):
        e.reset(slots=(np.arange(N) * 5) % 64)
        e.set_policy_seed(77, 0)
        if policy == 'ring':
            g = torch.Generator(device='cuda'); g.manual_seed(3)
            ring = torch.rand((7, N, n), device='cuda', generator=g, dtype=torch.float32)
            o = e.rollout(actions=ring, steps=300)
        else:
            o = e.rollout(policy=policy, steps=300)
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in o.items() if v is not None})
    for key in ('obs', 'reward', 'terminated', 'returns'):
        assert torch.equal(outs[0][key], outs[1][key]), key
    s0, s1 = fast.get_state(), general.get_state()
    for key in s0:
        assert np.array_equal(s0[key], s1[key]), key
    fast.close(); general.close()

