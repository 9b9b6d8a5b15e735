"""The certifier of tests/kkt.py, checked on the CPU: it accepts optima known in closed form (one pod breaker: water-filling
with the multiplier found by bisection in numpy; a feasible target: itself), accepts what SciPy's SLSQP finds to SLSQP's
accuracy, accepts the oracle's projections to the oracle's 2^-16 A snap — and REJECTS feasible points that are not the
optimum (mass moved between two free stations of a full pod; a point scaled inside the limits) and infeasible ones."""
import numpy as np
import pytest

import kkt


@pytest.fixture(scope='module')
def problem(caltech):
    return caltech, kkt.a_tilde(caltech.constraint_matrix, caltech.phase_angles)


def _pod_waterfill(b, h, cap):
    lo, hi = 0.0, 64.0
    for _ in range(200):
        nu = 0.5 * (lo + hi)
        if np.clip(b - nu, 0, h).sum() > cap:
            lo = nu
        else:
            hi = nu
    return np.clip(b - 0.5 * (lo + hi), 0, h)


def _pod_instance(net, rng):
    n = net.num_stations
    cc = np.flatnonzero(np.asarray(net.evse_kind) == 1)                   # the Caltech CC pod: 8 stations behind an 80 A breaker
    b = np.zeros(n); h = np.zeros(n)
    b[cc] = rng.uniform(4, 32, len(cc)); h[cc] = rng.uniform(8, 32, len(cc))
    y = np.clip(b, 0, h)
    y[cc] = _pod_waterfill(b[cc], h[cc], 80.0)
    return b, h, y, cc


def test_accepts_closed_form_optima_and_rejects_neighbours(problem):
    net, At = problem
    rng = np.random.default_rng(0)
    B, H, Y, Ybad, Yscaled = [], [], [], [], []
    for _ in range(64):
        b, h, y, cc = _pod_instance(net, rng)
        if np.clip(b, 0, h)[cc].sum() <= 80.0:
            continue
        free = [i for i in cc if 1e-3 < y[i] < h[i] - 1e-3]
        if len(free) < 2:
            continue
        bad = y.copy(); bad[free[0]] += 1e-4; bad[free[1]] -= 1e-4         # same pod sum: still feasible, no longer the projection
        B.append(b); H.append(h); Y.append(y); Ybad.append(bad); Yscaled.append(y * 0.999)
    assert len(B) > 20
    B, H, Y, Ybad, Yscaled = map(np.array, (B, H, Y, Ybad, Yscaled))
    ok = kkt.certify(At, net.magnitudes, B, H, Y)
    assert ok.moved.all() and (ok.n_active == 1).all()
    assert ok.stationarity.max() < 1e-11 and ok.row_excess.max() < 1e-13 and ok.min_lambda.min() > 0
    bad = kkt.certify(At, net.magnitudes, B, H, Ybad)
    assert bad.row_excess.max() < 1e-13 and bad.stationarity.min() > 0.9e-4      # feasible, flagged
    scaled = kkt.certify(At, net.magnitudes, B, H, Yscaled)                     # strictly inside every row: b - y must vanish on free coordinates
    assert scaled.stationarity.min() > 1e-3
    over = kkt.certify(At, net.magnitudes, B, H, Y * 1.001)
    assert over.row_excess.min() > 0.9e-3


def test_feasible_target_is_its_own_projection(problem):
    net, At = problem
    rng = np.random.default_rng(1)
    b = rng.uniform(0, 3, (32, net.num_stations)); h = np.full_like(b, 32.0)
    c = kkt.certify(At, net.magnitudes, b, h, b.copy())
    assert not c.moved.any() and c.stationarity.max() == 0.0 and c.row_excess.max() < 0


def test_accepts_scipy_and_oracle_projections_to_their_accuracy(problem):
    from oracle import binding as ob
    net, At = problem
    onet = ob.OracleNetwork(net)
    rng = np.random.default_rng(2)
    n = net.num_stations
    worst_scipy = worst_oracle = 0.0
    rows_seen = set()
    for trial in range(24):
        dem = np.where(rng.random(n) < 0.8, rng.uniform(0.1, 40, n), 0).astype(np.float32)
        a = rng.uniform(0, 1, n) if trial % 3 else np.ones(n)
        b, h = a * 32.0, kkt.upper_bound_amps(dem)
        x, rc, _ = onet.project(a, dem)
        assert rc == 0
        c = kkt.certify(At, net.magnitudes, b, h, x * 32.0, active_rtol=1e-5, box_atol=2.0 ** -16, accept=1e-5)
        worst_oracle = max(worst_oracle, float(c.stationarity[0]))
        rows_seen.add(int(c.n_active[0]))
        assert c.row_excess[0] <= n * 2.0 ** -17 / net.magnitudes.min()
        if trial < 8:
            ys = kkt.scipy_projection(At, net.magnitudes, b, h)
            cs = kkt.certify(At, net.magnitudes, b, h, ys, active_rtol=1e-5, box_atol=2.0 ** -16, accept=1e-5)
            worst_scipy = max(worst_scipy, float(cs.stationarity[0]))
            assert np.max(np.abs(ys - x * 32.0)) < 1e-3
    assert worst_oracle <= 2.0 ** -16 and worst_scipy < 1e-3
    assert max(rows_seen) >= 2                                                   # several rows at their limit at once were seen
