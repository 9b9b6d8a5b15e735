"""A synthetic network at the limits of the descriptor (n = 64 stations, 14 station classes -> 7 packed
words, m = 16 constraint rows, mixed AV / CC EVSEs) through every kernel path: the streaming kernel
of both layouts with and without debug outputs, the in-row water-filling (class-cap rows) and the slow
kernel (multi-class rows), against the oracle."""
import os

import numpy as np
import pytest

from helpers import assert_step_parity, make_workload, random_network
from sustaingym_amd.network import ChargingNetwork

pytestmark = pytest.mark.gpu


def big_network():
    n, G = 64, 14
    rng = np.random.default_rng(4)
    sizes = [5, 4, 6, 3, 5, 4, 6, 4, 5, 4, 5, 4, 5, 4]
    assert sum(sizes) == n
    cls = np.repeat(np.arange(G), sizes)
    phase_of = np.array([30.0, -90.0, 150.0])[np.arange(G) % 3]
    rows, mags, names = [], [], []
    for g in range(6):                                   # six class caps ("pods"): simple rows
        rows.append((cls == g).astype(float)); mags.append(20.0 * sizes[g] * 0.45); names.append(f'pod{g}')
    for r in range(10):                                  # ten feeder-like rows over several classes
        coef = np.zeros(G)
        pick = rng.choice(G, size=4, replace=False)
        coef[pick] = rng.choice([1.0, -1.0, 0.5], size=4)
        rows.append(coef[cls]); mags.append(float(rng.uniform(110, 220))); names.append(f'feeder{r}')
    return ChargingNetwork(site='synthetic64', station_ids=[f'S{i:02d}' for i in range(n)],
                           constraint_matrix=np.array(rows), phase_angles=phase_of[cls], magnitudes=np.array(mags),
                           constraint_names=names, evse_kind=(np.arange(n) % 5 == 0).astype(np.uint8))


@pytest.mark.parametrize('layout', ['compact', 'dense'])
@pytest.mark.parametrize('project', [True, False])
def test_synthetic_64_station_network(layout, project):
    from helpers import make_pair
    net = big_network()
    N = 96
    wl = make_workload(net, N, seed=8, busy=True, stride=96)
    old = os.environ.get('EVC_LAYOUT')
    os.environ['EVC_LAYOUT'] = layout
    try:
        eng, ob = make_pair(net, N, wl, project=project, debug=True)
        lean, _ = make_pair(net, N, wl, project=project, debug=False)
    finally:
        if old is None:
            del os.environ['EVC_LAYOUT']
        else:
            os.environ['EVC_LAYOUT'] = old
    g_obs = eng.reset(host=True).copy()
    lean.reset(host=True)
    assert np.array_equal(g_obs, ob.reset())
    rng = np.random.default_rng(1)
    slow = 0
    for t in range(288):
        a = rng.random((N, 64), dtype=np.float32)
        g = eng.step(a)
        o = ob.step(a)
        assert_step_parity(g, o, 64, tag=f'{layout} t={t}')
        l = lean.step(a)                                   # the kernels without debug outputs agree too
        assert np.array_equal(l['terminated'], g['terminated'])
        np.testing.assert_allclose(l['obs'], g['obs'], rtol=0, atol=2e-5)
        np.testing.assert_allclose(l['reward'], g['reward'], rtol=1e-11, atol=1e-13)
        if project:
            slow += eng.last_slow_count()
    if project:
        assert slow > 0                                    # the slow kernel took part
    eng.close(); lean.close()


def test_caps_violated_beside_multi_class_rows():
    """Found by tests/soak/network_fuzz.py (seed 0, case 1): six single-station classes, each with a cap row,
    plus nine rows over several classes.  Box-clipped actions violate caps AND multi-class rows at once while
    the projection only needs the caps; the slow kernel used to hand this to the conic-dual Newton (more
    violated rows than it keeps active) and raised EVC_STATUS_PROJ_NOCONV with a wrong schedule.  Caps are now
    filled first everywhere (relaxation argument), the rest only if rows stay violated."""
    from helpers import make_pair
    rng = np.random.default_rng(0)
    random_network(rng)                                   # case 0 of the fuzz sequence
    net = random_network(rng, 'fuzz1')
    n = net.num_stations
    assert n == 6 and len(net.magnitudes) == 15
    N = 64
    wl = make_workload(net, N, seed=101, busy=True, stride=96)
    eng, ob = make_pair(net, N, wl, project=True, debug=True)
    assert np.array_equal(eng.reset(host=True), ob.reset())
    arng = np.random.default_rng(1)
    for t in range(288):
        a = arng.random((N, n), dtype=np.float32)
        if t % 50 == 25:
            a[::3] = 1.0
        assert_step_parity(eng.step(a), ob.step(a), n, tag=f't={t}')
    assert not (eng.env_scalars()['status'] & 2).any()
    eng.close()
