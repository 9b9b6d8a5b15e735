"""The two state layouts of the engine: station rows (EVC_LAYOUT=dense) and entry lists
(compact, the default).  Same C-ABI, same results: side-by-side runs through the lean (no debug
outputs) kernels on a quiet and on a congested network, with and without projection, across an
autoreset boundary; the congested case has > 16 and > 32 EVs plugged in (all four entry slots)."""
import os

import numpy as np
import pytest

from helpers import make_workload
from sustaingym_amd.network import caltech_acn, jpl_acn

pytestmark = pytest.mark.gpu


def _engine(layout, net, N, wl, project, debug):
    from sustaingym_amd.engine import StepEngine
    old = os.environ.get('EVC_LAYOUT')
    os.environ['EVC_LAYOUT'] = layout
    try:
        eng = StepEngine(net, N, project_action=project, autoreset=True, bank_slots=len(wl['n_sessions']),
                         max_sessions=wl['sessions'].shape[1], moer_days=wl['moer'].shape[0], debug_outputs=debug)
    finally:
        if old is None:
            del os.environ['EVC_LAYOUT']
        else:
            os.environ['EVC_LAYOUT'] = old
    eng.upload_moer(wl['moer'], 0)
    eng.upload_episodes(wl['n_sessions'], wl['sessions'], wl['requested'], wl['moer_day'], 0)
    eng.set_autoreset_stride(1)
    return eng


@pytest.mark.parametrize('site,busy,project', [('caltech', False, True), ('caltech', True, True),
                                               ('caltech', True, False), ('jpl', True, True)])
def test_layouts_agree_step_for_step(site, busy, project):
    net = caltech_acn() if site == 'caltech' else jpl_acn()
    N, n = 203, net.num_stations                 # odd: a partial quad at the end
    wl = make_workload(net, N, bank_slots=N + 5, seed=31, busy=busy, stride=96 if busy else 64)
    dense = _engine('dense', net, N, wl, project, debug=False)
    comp = _engine('compact', net, N, wl, project, debug=False)
    d_obs = dense.reset(host=True).copy()
    c_obs = comp.reset(host=True).copy()
    assert np.array_equal(d_obs, c_obs)
    rng = np.random.default_rng(3)
    peak = 0
    for t in range(300):                         # crosses the episode boundary at 288
        a = rng.random((N, n), dtype=np.float32)
        if t % 50 == 7:
            a[::9] = 1.0                          # saturated rows: pods and feeders bind
        d = dense.step(a)
        c = comp.step(a)
        assert np.array_equal(d['terminated'], c['terminated']), t
        assert np.array_equal(d['obs'][:, n:2 * n], c['obs'][:, n:2 * n]), t      # est_departures: integers
        np.testing.assert_allclose(c['obs'], d['obs'], rtol=0, atol=2e-5, err_msg=f't={t}')
        np.testing.assert_allclose(c['reward'], d['reward'], rtol=1e-11, atol=1e-13, err_msg=f't={t}')
        np.testing.assert_allclose(c['breakdown'], d['breakdown'], rtol=1e-11, atol=1e-12, err_msg=f't={t}')
        if t == 287:
            assert d['terminated'].all()
            np.testing.assert_allclose(c['final_obs'], d['final_obs'], rtol=0, atol=2e-5)
        peak = max(peak, int((c['obs'][:, :n] > 0).sum(axis=1).max()))
    if busy:
        assert peak > 32, peak                   # all four entry slots were in use
    # the station view of the state is the same under both layouts
    for x, y in zip(dense.station_state(), comp.station_state()):
        if x.dtype.kind == 'f':
            np.testing.assert_allclose(y, x, rtol=1e-11, atol=1e-12)
        else:
            assert np.array_equal(x, y)
    ds, cs = dense.env_scalars(), comp.env_scalars()
    for key in ds:
        assert np.array_equal(ds[key], cs[key]), key
    dense.close(); comp.close()


def test_status_word_and_entry_count_do_not_interfere():
    """Compact layout keeps the entry count in the upper bits of the status word: host accessors must not
    see it, clear_status / set_env_scalars must not destroy it, and stepping past the end of an episode
    without autoreset must leave the entries alone."""
    from sustaingym_amd import _lib
    from sustaingym_amd.engine import StepEngine
    net = caltech_acn()
    N, n = 64, net.num_stations
    wl = make_workload(net, N, seed=5, busy=True, stride=96)
    os.environ['EVC_LAYOUT'] = 'compact'
    try:
        eng = StepEngine(net, N, project_action=True, autoreset=False, bank_slots=N, max_sessions=96, moer_days=3)
    finally:
        del os.environ['EVC_LAYOUT']
    eng.upload_moer(wl['moer'], 0)
    eng.upload_episodes(wl['n_sessions'], wl['sessions'], wl['requested'], wl['moer_day'], 0)
    eng.reset(host=True)
    rng = np.random.default_rng(0)
    for t in range(150):
        a = rng.random((N, n), dtype=np.float32)
        if t == 100:
            a[3, 5] = 1.5                              # out of range -> ACTION_CLAMPED on env 3
        eng.step(a)
    rem0, dep0, est0 = eng.station_state()
    assert (dep0 >= 0).sum(axis=1).max() > 16          # entry counts well above one slot
    sc = eng.env_scalars()
    assert sc['status'][3] & _lib.STATUS_ACTION_CLAMPED and sc['status'].max() < 16     # flags only
    _lib.check(eng.lib.evc_clear_status(eng.handle), 'evc_clear_status')
    assert not eng.env_scalars()['status'].any()
    rem1, dep1, est1 = eng.station_state()             # entries survived the status rewrite
    assert np.array_equal(dep0, dep1) and np.array_equal(est0, est1) and np.array_equal(rem0, rem1)
    out = eng.step(rng.random((N, n), dtype=np.float32))
    assert not out['terminated'].any()
    for t in range(151, 288):
        out = eng.step(rng.random((N, n), dtype=np.float32))
    assert out['terminated'].all()
    rem2, dep2, est2 = eng.station_state()
    out = eng.step(rng.random((N, n), dtype=np.float32))          # step after termination: no-op + flag
    assert out['terminated'].all() and not out['reward'].any()
    assert (eng.env_scalars()['status'] & _lib.STATUS_STEP_AFTER_DONE).all()
    rem3, dep3, est3 = eng.station_state()
    assert np.array_equal(dep2, dep3) and np.array_equal(rem2, rem3)
    eng.close()


@pytest.mark.parametrize('k', [1, 12])
@pytest.mark.parametrize('layout', ['compact', 'dense'])
def test_short_moer_forecasts(k, layout):
    """moer_forecast_steps below the default 36 (env.py:120 allows 1..36): observation width 2n + k + 2,
    forecast columns 1..k of the MOER matrix; against the oracle, debug and production kernels."""
    from helpers import assert_step_parity, make_pair
    net = caltech_acn()
    N, n = 40, net.num_stations
    wl = make_workload(net, N, seed=2)
    old = os.environ.get('EVC_LAYOUT')
    os.environ['EVC_LAYOUT'] = layout
    try:
        eng, ob = make_pair(net, N, wl, project=True, autoreset=True, k=k, debug=True)
        lean, _ = make_pair(net, N, wl, project=True, autoreset=True, k=k, debug=False)
    finally:
        if old is None:
            del os.environ['EVC_LAYOUT']
        else:
            os.environ['EVC_LAYOUT'] = old
    assert eng.F == 2 * n + k + 2
    assert np.array_equal(eng.reset(host=True), ob.reset())
    lean.reset(host=True)
    rng = np.random.default_rng(k)
    for t in range(300):
        a = rng.random((N, n), dtype=np.float32)
        g, o, l = eng.step(a), ob.step(a, autoreset=True), lean.step(a)
        assert g['obs'].shape == (N, 2 * n + k + 2)
        assert_step_parity(g, o, n, tag=f'k={k} t={t}')
        assert np.array_equal(l['obs'], g['obs']) and np.array_equal(l['terminated'], g['terminated'])
    eng.close(); lean.close()
