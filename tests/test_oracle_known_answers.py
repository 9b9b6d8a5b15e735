"""Known-answer tests K1-K7 of SURVEY.md §8c for the CPU oracle (hand-derived from the
reference's formulas, env.py:99-114,340-394,431-464 and the published acnportal battery /
event semantics).  These pin the oracle; the GPU parity tests then pin the HIP engine to it."""
import numpy as np
import pytest

from oracle import binding as ob
from sustaingym_amd.network import caltech_acn

A_PERS_TO_KWH = (1 / 60) * (208 / 1000) * 5
PROFIT_FACTOR = A_PERS_TO_KWH * (0.15 * 0.20)
CARBON_COST_FACTOR = A_PERS_TO_KWH * (30.85 / 1000)
VIOLATION_FACTOR = A_PERS_TO_KWH * 0.001


@pytest.fixture(scope='module')
def net():
    return caltech_acn()


@pytest.fixture(scope='module')
def onet(net):
    return ob.OracleNetwork(net)


def flat_moer(value=0.3):
    m = np.full((289, 37), value)
    m[:, 0] = value + 0.001 * np.arange(289) / 289
    return m


def make_env(onet, sessions, requested, project=False, moer=None, k=36, charge_calculation='continuous'):
    env = ob.OracleEnv(onet, k, project, charge_calculation)
    s = ob.pack_sessions(*zip(*sessions)) if sessions else np.zeros(0, ob.SESSION_DTYPE)
    obs = env.reset(s, np.array(requested, dtype=np.float64), flat_moer() if moer is None else moer)
    return env, obs


def test_constants():
    assert abs(A_PERS_TO_KWH - 0.017333333333333333) < 1e-15
    assert abs(PROFIT_FACTOR - 5.2e-4) < 1e-15
    assert abs(VIOLATION_FACTOR - 1.7333333333e-5) < 1e-12
    assert abs(CARBON_COST_FACTOR - 5.347333333e-4) < 1e-12


def test_k1_single_ev_full_rate(onet, net):
    """K1: one EV, 30 kWh, 32 A: 6.656 kW -> 0.554667 kWh/step, reward = 32*(5.2e-4 - 5.347e-4*moer)."""
    st = 3
    env, obs = make_env(onet, [(0, 100, 90, st)], [30.0])
    n = net.num_stations
    assert np.all(obs[:2 * n] == 0)          # nothing plugged at t=0 (events at ts 0 pop after 1st pass)
    a = np.zeros(n, np.float32)
    a[st] = 1.0
    obs, res = env.step(a)                    # iteration 0: EVSE empty, then EV plugs at it=1
    assert res.reward == 0.0 and res.rates[st] == 0.0
    assert obs[st] == np.float32(30.0) and obs[n + st] == 90 - 1
    obs, res = env.step(a)
    assert res.pilots[st] == 32.0
    assert abs(res.rates[st] - 32.0) < 1e-12
    moer = flat_moer()[2, 0]
    assert abs(res.reward - 32.0 * (PROFIT_FACTOR - CARBON_COST_FACTOR * moer)) < 1e-15
    assert abs(float(obs[st]) - (30.0 - 32 * A_PERS_TO_KWH)) < 1e-5
    assert abs(32 * A_PERS_TO_KWH - 0.5546666667) < 1e-9


def test_k2_taper_stepwise(onet, net):
    """K2 (legacy `_charge_stepwise`): remaining 1.0 kWh at soc >= 0.8 with 32 A -> P = 5 kW, rem -> 0.583333."""
    st = 0
    env, _ = make_env(onet, [(0, 100, 90, st)], [1.0], charge_calculation='stepwise')
    a = np.zeros(net.num_stations, np.float32)
    a[st] = 1.0
    env.step(a)
    obs, res = env.step(a)
    assert abs(res.rates[st] - 5.0 * 1000 / 208) < 1e-9
    rem, _, _ = env.station_state()
    assert abs(rem[st] - (1.0 - 5.0 / 12)) < 1e-12


def _one_period(onet, net, requested, amps, charge_calculation='continuous', st=0):
    """Remaining demand and delivered rate after ONE charging period of a fresh EV at `amps` (AV EVSE)."""
    env, _ = make_env(onet, [(0, 100, 90, st)], [requested], charge_calculation=charge_calculation)
    a = np.zeros(net.num_stations, np.float32)
    a[st] = amps / 32.0
    env.step(a)                               # EV plugs in after the first charge pass
    _, res = env.step(a)
    assert res.pilots[st] == amps
    rem, _, _ = env.station_state()
    return rem[st], res.rates[st]


def test_k2_continuous_battery(onet, net):
    """K2' (acnportal's default `_charge`, hand-derived): with kw = 0.208 pilot the ramp-down starts at headroom
    rem_T = 0.2 kw; 32 A: kw = 6.656, rem_T = 1.3312, kw/12 = 0.554667.
      * tail (rem <= rem_T): rem' = rem exp(-5/12): 1.0 kWh -> 0.659241 (the legacy model gives 0.583333)
      * crossing (0 < rem - rem_T < kw/12): 1.5 kWh -> rem_T exp(-(kw/12 - (1.5 - rem_T))/rem_T) = 0.996224
      * constant-rate (rem - rem_T >= kw/12): 30 kWh -> 30 - 0.554667, delivered rate = pilot
      * the tail factor does not depend on the pilot: 16 A, rem_T = 0.6656: 0.5 kWh -> 0.5 exp(-5/12)"""
    e = np.exp(-5.0 / 12.0)
    assert abs(e - 0.6592406302) < 1e-10
    rem, amps = _one_period(onet, net, 1.0, 32.0)
    assert abs(rem - e) < 1e-12
    assert abs(amps - (1.0 - e) * 12 * 1000 / 208) < 1e-10
    rem, amps = _one_period(onet, net, 1.5, 32.0)
    kw, rem_t = 6.656, 1.3312
    want = rem_t * np.exp(-(kw / 12 - (1.5 - rem_t)) / rem_t)
    assert abs(want - 0.9962241555) < 1e-9
    assert abs(rem - want) < 1e-12
    assert abs(amps - (1.5 - want) * 12 * 1000 / 208) < 1e-10
    rem, amps = _one_period(onet, net, 30.0, 32.0)
    assert abs(rem - (30.0 - kw / 12)) < 1e-12 and abs(amps - 32.0) < 1e-11
    rem, amps = _one_period(onet, net, 0.5, 16.0)
    assert abs(rem - 0.5 * e) < 1e-12
    # the two regions meet continuously: just above / below rem_T + kw/12 and rem_T
    for r0 in (rem_t + kw / 12, rem_t):
        lo, _ = _one_period(onet, net, r0 - 1e-9, 32.0)
        hi, _ = _one_period(onet, net, r0 + 1e-9, 32.0)
        assert abs(hi - lo) < 1e-8
    # both models agree wherever the battery stays in the constant-rate region
    for req in (5.0, 30.0, 99.0):
        assert abs(_one_period(onet, net, req, 32.0)[0] - _one_period(onet, net, req, 32.0, 'stepwise')[0]) < 1e-12


def test_k2_continuous_session_never_overshoots(onet, net):
    """A whole session at full rate under the continuous model: demand decreases monotonically, stays >= 0,
    energy delivered never exceeds the request, and the EV leaves the observation once below 1e-3 kWh."""
    st = 0
    env, _ = make_env(onet, [(0, 200, 190, st)], [4.0])
    a = np.zeros(net.num_stations, np.float32)
    a[st] = 1.0
    env.step(a)
    last, gone = 4.0, None
    for i in range(60):
        obs, res = env.step(a)
        rem = env.station_state()[0][st]
        assert 0.0 <= rem <= last + 1e-15 and res.rates[st] >= 0.0
        last = rem
        if rem <= 1e-3 and gone is None:
            gone = i
            assert obs[st] == 0.0 and obs[net.num_stations + st] == 0.0     # EV.fully_charged -> not active
    assert gone is not None and gone > 13                                  # the legacy model needs 13 periods


def test_k3_rounding(onet, net):
    """K3: AV {5.99->0, 6.0->6, 6.5->6, 7.5->8, 31.5->32}; CC {3.9->0, 4.0->0, 12->16, 20->16, 28->32}."""
    env, _ = make_env(onet, [], [])
    av = [i for i in range(net.num_stations) if net.evse_kind[i] == 0][:5]
    cc = [i for i in range(net.num_stations) if net.evse_kind[i] == 1][:5]
    a = np.zeros(net.num_stations, np.float64)
    for i, amps in zip(av, (5.99, 6.0, 6.5, 7.5, 31.5)):
        a[i] = amps / 32
    for i, amps in zip(cc, (3.9, 4.0, 12.0, 20.0, 28.0)):
        a[i] = amps / 32
    _, res = env.step(a.astype(np.float32))
    got_av = [res.pilots[i] for i in av]
    got_cc = [res.pilots[i] for i in cc]
    # float32 actions: 5.99/32 and 31.5/32 etc. are re-derived in float32 exactly as the env would see them
    exp_av = []
    for amps in (5.99, 6.0, 6.5, 7.5, 31.5):
        y = float(np.float32(amps / 32)) * 32
        exp_av.append(float(np.round(y)) if y >= 6 else 0.0)
    assert got_av == exp_av == [0.0, 6.0, 6.0, 8.0, 32.0]
    assert got_cc == [0.0, 0.0, 16.0, 16.0, 32.0]


def test_k4_constraint_rows(onet, net):
    """K4: 8 CC-pod stations at 16 A = 128 A vs 80 A -> 48 A excess -> 8.32e-4 $."""
    env, _ = make_env(onet, [], [])
    a = np.zeros(net.num_stations, np.float32)
    cc = [i for i in range(net.num_stations) if net.evse_kind[i] == 1]
    assert len(cc) == 8
    a[cc] = 0.5
    _, res = env.step(a)
    # secondary / primary rows: |I3a| = 128 (AB only) etc. all below limits
    cur = np.abs(net.constraint_current(np.array(res.pilots[:net.num_stations])))
    exc = np.sum(np.maximum(0, cur - net.magnitudes))
    assert abs(exc - 48.0) < 1e-9
    assert abs(res.reward + 48.0 * VIOLATION_FACTOR) < 1e-15
    assert abs(48.0 * VIOLATION_FACTOR - 8.32e-4) < 1e-9


def test_k4b_primary_row_magnitude(net):
    """Phasor algebra of the delta network: |I3a| with S_ab=400, S_ca=200 = sqrt(400^2+200^2+400*200)."""
    sched = np.zeros(net.num_stations)
    ab = [i for i in range(net.num_stations) if net.phase_angles[i] == 30.0]
    ca = [i for i in range(net.num_stations) if net.phase_angles[i] == 150.0]
    sched[ab[:20]] = 20.0
    sched[ca[:10]] = 20.0
    cur = np.abs(net.constraint_current(sched))
    assert abs(cur[0] - np.sqrt(400 ** 2 + 200 ** 2 + 400 * 200)) < 1e-9


def test_k5_event_ordering(onet, net):
    """K5: arrival 0 plugs at iteration 1; unplug precedes plug-in at equal timestamps; the
    departure-before-next-pass rule; done exactly at step 288."""
    st = 5
    sessions = [(0, 3, 3, st), (3, 10, 9, st), (20, 20, 25, 7)]   # 3rd: departure bin == arrival bin
    env, obs = make_env(onet, sessions, [10.0, 12.0, 8.0])
    n = net.num_stations
    z = np.zeros(n, np.float32)
    status = 0
    for t in range(1, 289):
        obs, res = env.step(z)
        status |= res.status
        rem, dep, est = env.station_state()
        if t in (1, 2):
            assert dep[st] == 3 and obs[st] == np.float32(10.0)
        if t == 3:                       # EV0 unplugged, EV1 plugged in the same pass, no conflict
            assert dep[st] == 10 and obs[st] == np.float32(12.0) and obs[n + st] == 9 - 3
        if t == 10:
            assert dep[st] == -1
        if t == 20:                      # plugged at 20, unplug event (ts 20) only seen at the next pass
            assert dep[7] == 20 and obs[7] == np.float32(8.0)
        if t == 21:
            assert dep[7] == -1
        assert res.terminated == (1 if t == 288 else 0)
    assert status == 0
    assert abs(float(obs[2 * n + 36 + 1]) - 1.0) < 1e-7


def test_k5b_occupied_station_flag(onet, net):
    sessions = [(5, 50, 40, 2), (10, 60, 50, 2)]
    env, _ = make_env(onet, sessions, [10.0, 10.0])
    z = np.zeros(net.num_stations, np.float32)
    st = 0
    for _ in range(12):
        _, res = env.step(z)
        st |= res.status
    assert st & 1      # ORC_STATUS_OCCUPIED


def test_k6_fully_charged_disappears(onet, net):
    """K6: an EV whose remaining demand drops to <= 1e-3 kWh vanishes from the observation."""
    st = 1
    env, _ = make_env(onet, [(0, 200, 150, st)], [0.5])
    n = net.num_stations
    a = np.zeros(n, np.float32)
    a[st] = 1.0
    seen_zero = False
    for t in range(1, 40):
        obs, res = env.step(a)
        rem, dep, _ = env.station_state()
        assert dep[st] == 200
        if rem[st] <= 1e-3:
            assert obs[st] == 0.0 and obs[n + st] == 0.0
            seen_zero = True
        else:
            assert obs[st] > 0
    assert seen_zero


def test_k7_projection_feasible_returns_itself(onet, net):
    rng = np.random.default_rng(0)
    n = net.num_stations
    a = rng.uniform(0, 0.2, n)
    dem = np.full(n, 50.0, np.float32)
    x, rc, kkt = onet.project(a, dem)
    assert rc == 0 and np.array_equal(x, a)


def test_k7_projection_single_pod_closed_form(onet, net):
    """One active linear row: x = clip(a - lambda) on the pod (water-filling)."""
    n = net.num_stations
    cc = [i for i in range(n) if net.evse_kind[i] == 1]
    a = np.zeros(n)
    a[cc] = np.array([1.0, 0.9, 0.8, 0.1, 0.0, 0.0, 0.0, 0.0])
    dem = np.full(n, 50.0, np.float32)
    x, rc, kkt = onet.project(a, dem)
    assert rc == 0
    # 32*(1 + .9 + .8 + .1) = 89.6 A > 80: shift lambda with the 0.1 station clamped at 0?
    # try both hypotheses in closed form
    y = a[cc] * 32
    lam = (y[:4].sum() - 80) / 4
    if y[3] - lam < 0:
        lam = (y[:3].sum() - 80) / 3
    exp = np.clip(y - lam, 0, 32)
    assert np.allclose(x[cc] * 32, exp, atol=1e-8)
    assert abs(exp.sum() - 80) < 1e-9
    assert np.all(x[[i for i in range(n) if i not in cc]] == 0)


def test_k7_projection_kkt_and_scipy(onet, net):
    from scipy.optimize import minimize
    rng = np.random.default_rng(1)
    n = net.num_stations
    At = net.a_tilde()
    r = net.magnitudes
    worst = 0.0
    for trial in range(40):
        occ = rng.random(n) < (0.6 + 0.4 * (trial % 2))
        dem = np.where(occ, rng.uniform(0.1, 40, n), 0).astype(np.float32)
        a = rng.uniform(0, 1, n) if trial % 3 else np.ones(n)
        x, rc, kkt = onet.project(a, dem)
        assert rc == 0
        # KKT certificate of the converged point (before the 2^-16 A tie snap)
        assert kkt[0] < 1e-9 and kkt[1] < 1e-10 and kkt[2] < 1e-8 and kkt[3] < 1e-8
        u = np.minimum(1.0, dem.astype(np.float64) / A_PERS_TO_KWH / 32)
        assert np.all(x >= 0) and np.all(x <= u)
        # the returned point is the optimum snapped to a 2^-16 A grid: feasible to n * 2^-17 A
        assert np.all(np.abs(At @ x) * 32 <= r + n * 2.0 ** -17)
        if trial < 12:
            cons = [{'type': 'ineq', 'fun': (lambda v, c=c: r[c] ** 2 - (np.abs(At[c] @ v) * 32) ** 2)}
                    for c in range(len(r))]
            ref = minimize(lambda v: np.sum((v - a) ** 2), np.minimum(a, u) * 0.5, jac=lambda v: 2 * (v - a),
                           bounds=list(zip(np.zeros(n), u)), constraints=cons, method='SLSQP',
                           options={'ftol': 1e-15, 'maxiter': 400})
            worst = max(worst, np.max(np.abs(ref.x - x)))
            # our point is feasible and at least as close to `a` as SciPy's
            assert np.sum((x - a) ** 2) <= np.sum((ref.x - a) ** 2) + 1e-5  # snap + SLSQP slack
    assert worst < 1e-4


def test_max_profit_formula():
    s = ob.pack_sessions([0, 10], [100, 12], [90, 20], [0, 1])
    req = np.array([30.0, 50.0])
    exp = (min(30.0, 100 * 32 * A_PERS_TO_KWH) + min(50.0, 2 * 32 * A_PERS_TO_KWH)) * 0.03
    assert abs(ob.max_profit(s, req) - exp) < 1e-12


def test_discrete_wrapper_mapping(onet, net):
    env, _ = make_env(onet, [], [])
    n = net.num_stations
    act = np.arange(n) % 5
    _, res = env.step_discrete(act, 5)
    for i in range(n):
        y = (act[i] / 4) * 32
        exp = (np.round(y) if y >= 6 else 0.0) if net.evse_kind[i] == 0 else np.round(y / 8) * 8
        assert res.pilots[i] == exp
