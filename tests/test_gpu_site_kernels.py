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


@pytest.mark.parametrize('busy', [False, True])
@pytest.mark.parametrize('project', [True, False])
@pytest.mark.parametrize('site', ['caltech', 'jpl'])
def test_site_kernels_equal_the_general_kernels(site, project, busy, caltech, jpl, monkeypatch):
    net = caltech if site == 'caltech' else jpl
    N, n = 1000, net.num_stations                   # not a multiple of 4: a ragged last quad
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
