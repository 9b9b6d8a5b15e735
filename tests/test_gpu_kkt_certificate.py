"""a3 (`_project_action`, env.py:178-221 + `magnitude_constraint`, env.py:473-500): the HIP kernels' OWN projected vector is
certified as the optimum of the reference's optimisation problem — feasibility + KKT stationarity with non-negative
multipliers, checked in numpy / SciPy by ``tests/kkt.py`` — with no call into ``oracle/`` and no solver of this repository
in the loop.  MOSEK computes that same optimum to its tolerance, so this is parity with the reference's a3 up to solver
accuracy, independent of the oracle (VERDICT r4 "next" #1).

How the un-snapped optimum is read out of the PRODUCT kernels: ``evc_set_tie_grid(e, 40)`` moves the tie-snap grid from
2^-16 A to 2^-40 A (~1e-12 A), a constant of the same code path; ``projected`` of the per-station debug outputs is then the
solvers' output to its last bits.  A second engine WITHOUT debug outputs (the lean streaming kernels + in-kernel drain that
bench.py times) runs in lock-step on the same bank and actions and must give bitwise equal observations, rewards and event
state: its projections are the certified ones.  The reach of the default 2^-16 A snap is measured beside it: the fraction of
solver-moved values whose pilot (env.py:373-378) differs between the un-snapped and the snapped value.
"""
import json
import os

import numpy as np
import pytest

import kkt
from helpers import make_workload, random_network
from sustaingym_amd.hostio import to_device, to_host
from test_gpu_rollout import _gmm_engine

pytestmark = pytest.mark.gpu

N, BANK = 8192, 8192
ROW_TOL = 1.5e-10          # the kernels accept a row at |A~ y| <= magnitude (1 + 1e-10) (Consts::PROJ_TOL)
STAT_TOL = 1e-8            # amps, max-norm of the part of 32 (a - x) outside the normal cone
REPORT = {}


def _families(net, rng, n_envs):
    n = net.num_stations
    fam = {'uniform': (rng.random((n_envs, n), dtype=np.float32), 0), 'ones': (np.ones((n_envs, n), np.float32), 0)}
    for bins in (2, 3, 5, 9):
        fam[f'bins{bins}'] = (rng.integers(0, bins, (n_envs, n), dtype=np.int64), bins)
    A = np.abs(net.constraint_matrix)
    tie = rng.random((n_envs, n), dtype=np.float32)
    for c in range(A.shape[0]):                                  # every station of a pod at cap / size: the row sits ON its limit
        cols = np.flatnonzero(A[c] > 0)
        if len(np.unique(net.phase_angles[cols])) == 1 and len(cols) <= 16:
            tie[:, cols] = np.float32(net.magnitudes[c] / A[c, cols[0]] / len(cols) / 32.0)
    fam['pod_tie'] = (tie, 0)
    fam['two_level'] = (np.where(rng.random((n_envs, n)) < 0.5, np.float32(0.25), np.float32(1.0)).astype(np.float32), 0)
    fam['greedy'] = (None, 0)                                    # sign(demands) of the previous observation (baselines.py:22-35)
    return fam


def _as_float(a, bins):
    return a if bins == 0 else (a.astype(np.float32) / np.float32(bins - 1))          # wrappers.py:43-45


def _legal_pilot(y, is_cc):
    """env.py:373-378 on amps."""
    return np.where(is_cc, np.round(y / 8) * 8, np.where(y >= 6, np.round(y), 0.0))


def _tie_snap(y, h, k=16):
    """DESIGN.md §4 (the one deliberate deviation): nearest point of the 2^-k A grid offset by sqrt(2)-1 steps."""
    off = 0.41421356237309515
    return np.clip(np.ldexp(np.rint(np.ldexp(y, k) - off) + off, -k), 0.0, h)


class _Tally:
    def __init__(self):
        self.worst = {'row_excess': -1.0, 'box_excess': -1.0, 'stationarity': 0.0, 'complementarity': 0.0, 'min_lambda': 0.0}
        self.instances = self.moved = self.active = self.nnls = 0
        self.moved_values = self.snap_flips = 0
        self.by_rows = {}

    def add(self, cert, y, h, b, is_cc):
        w = cert.worst()
        for k in ('row_excess', 'box_excess', 'stationarity', 'complementarity'):
            self.worst[k] = max(self.worst[k], w[k])
        self.worst['min_lambda'] = min(self.worst['min_lambda'], w['min_lambda'])
        self.instances += w['instances']; self.moved += w['moved']; self.active += w['with_active_rows']; self.nnls += w['nnls']
        for k, c in zip(*np.unique(cert.n_active[cert.moved], return_counts=True)):
            self.by_rows[int(k)] = self.by_rows.get(int(k), 0) + int(c)
        mv = y != np.minimum(b, h)                                # values a solver moved: the ones the product snaps
        self.moved_values += int(mv.sum())
        cc = np.broadcast_to(is_cc, y.shape)
        self.snap_flips += int((_legal_pilot(y, cc) != _legal_pilot(_tie_snap(y, h), cc))[mv].sum())

    def record(self, name):
        REPORT[name] = dict(self.worst, instances=self.instances, congested=self.moved, with_active_rows=self.active,
                            nnls_fallbacks=self.nnls, active_rows_histogram=self.by_rows, solver_moved_values=self.moved_values,
                            pilots_changed_by_default_snap=self.snap_flips,
                            snap_flip_fraction=self.snap_flips / max(1, self.moved_values))
        os.makedirs('gpurun_out', exist_ok=True)
        with open('gpurun_out/kkt_certificate.json', 'w') as f:
            json.dump(REPORT, f, indent=1)
        print(name, json.dumps(REPORT[name]))

    def check(self, min_congested):
        assert self.worst['row_excess'] <= ROW_TOL, self.worst
        assert self.worst['box_excess'] <= 1e-12, self.worst
        assert self.worst['stationarity'] <= STAT_TOL, self.worst
        assert self.worst['min_lambda'] >= -STAT_TOL, self.worst
        assert self.moved >= min_congested, (self.moved, min_congested)


def _lockstep(net, dbg, lean, tally, t0, t1, seed):
    """Steps both engines t0 .. t1 with rotating action families, certifying the debug engine's projected vector."""
    n = net.num_stations
    At = kkt.a_tilde(net.constraint_matrix, net.phase_angles)
    is_cc = np.asarray(net.evse_kind) == 1
    rng = np.random.default_rng(seed)
    fam = _families(net, rng, dbg.N)
    names = list(fam)
    obs = to_host(dbg.reset()).copy()
    assert np.array_equal(obs, to_host(lean.reset()))
    u = rng.random((dbg.N, n), dtype=np.float32)
    for t in range(t1):
        name = 'uniform' if t < t0 else names[t % len(names)]
        a, bins = (u, 0) if t < t0 else fam[name]
        if a is None:
            a = np.sign(obs[:, :n]).astype(np.float32)
        g = {k: to_host(v).copy() for k, v in dbg.step(to_device(a), bins=bins).items()}
        l = {k: to_host(v) for k, v in lean.step(to_device(a), bins=bins).items()}
        for key in ('obs', 'reward', 'terminated', 'breakdown'):
            assert np.array_equal(g[key], l[key]), (name, t, key)
        if t >= t0 or t % 8 == 0:
            b = _as_float(a, bins).astype(np.float64) * 32.0
            h = kkt.upper_bound_amps(obs[:, :n])
            y = g['projected'] * 32.0
            cert = kkt.certify(At, net.magnitudes, b, h, y)
            bad = np.flatnonzero((cert.stationarity > STAT_TOL) | (cert.row_excess > ROW_TOL) | (cert.min_lambda < -STAT_TOL))
            assert len(bad) == 0, (name, t, bad[:5], cert.stationarity[bad[:5]], cert.row_excess[bad[:5]], cert.min_lambda[bad[:5]],
                                   cert.n_active[bad[:5]])
            tally.add(cert, y, h, b, is_cc)
        obs = g['obs']
    assert not (dbg.env_scalars()['status'] & 2).any()          # EVC_STATUS_PROJ_NOCONV never
    assert not (lean.env_scalars()['status'] & 2).any()


@pytest.mark.parametrize('site', ['caltech', 'jpl'])
def test_hip_projection_is_the_optimum_of_the_reference_problem_on_gmm_middays(site):
    period = 'Summer 2019' if site == 'caltech' else 'Summer 2021'
    net, dbg = _gmm_engine(site, period, N, BANK, seed=404, debug_outputs=True)
    _, lean = _gmm_engine(site, period, N, BANK, seed=404)
    dbg.set_tie_grid(40); lean.set_tie_grid(40)
    tally = _Tally()
    _lockstep(net, dbg, lean, tally, t0=96, t1=216, seed=21)
    tally.record(f'gmm_{site}')
    # first run (profiles/r5_kkt_certificate.json): 172 133 congested instances at Caltech, 35 921 on the provisional JPL set
    tally.check(min_congested=100_000 if site == 'caltech' else 25_000)
    dbg.close(); lean.close()


def test_hip_projection_is_the_optimum_on_random_networks():
    """helpers.random_network descriptors (3 .. 64 stations, 1 .. 14 classes, up to 16 rows with mixed signs), busy synthetic days."""
    from sustaingym_amd.engine import StepEngine
    tally = _Tally()
    for seed in range(12):
        net = random_network(np.random.default_rng(1000 + seed), tag=f'kkt{seed}')
        wl = make_workload(net, 2048, seed=seed, busy=True)
        engines = []
        for debug in (True, False):
            eng = StepEngine(net, 2048, project_action=True, bank_slots=2048, max_sessions=wl['sessions'].shape[1],
                             moer_days=wl['moer'].shape[0], debug_outputs=debug)
            eng.upload_moer(wl['moer'])
            eng.upload_episodes(wl['n_sessions'], wl['sessions'], wl['requested'], wl['moer_day'])
            eng.set_tie_grid(40)
            engines.append(eng)
        _lockstep(net, engines[0], engines[1], tally, t0=60, t1=132, seed=50 + seed)
        for eng in engines:
            eng.close()
    tally.record('random_networks')
    tally.check(min_congested=100_000)


def test_certifier_sees_the_default_snap_and_bounds_it():
    """At the default 2^-16 A grid the same kernels' output is the certified optimum moved by at most 2^-17 A per value: the
    stationarity residual is bounded by that, and feasibility holds to the slack the engine documents (n 2^-17 A per row)."""
    net, dbg = _gmm_engine('caltech', 'Summer 2019', 2048, 2048, seed=405, debug_outputs=True)
    n = net.num_stations
    At = kkt.a_tilde(net.constraint_matrix, net.phase_angles)
    rng = np.random.default_rng(3)
    obs = to_host(dbg.reset()).copy()
    worst = 0.0
    congested = 0
    for t in range(160):
        a = rng.random((2048, n), dtype=np.float32) if t % 2 else np.ones((2048, n), np.float32)
        g = {k: to_host(v).copy() for k, v in dbg.step(to_device(a)).items()}
        if t >= 100:
            y = g['projected'] * 32.0
            h = kkt.upper_bound_amps(obs[:, :n])
            cert = kkt.certify(At, net.magnitudes, a.astype(np.float64) * 32.0, h, y, active_rtol=1e-5, box_atol=2.0 ** -16, accept=1e-5)
            worst = max(worst, float(cert.stationarity.max()))
            congested += int(cert.moved.sum())
            assert np.all(np.abs(y @ At.T) <= net.magnitudes * (1 + 1e-10) + n * 2.0 ** -17)
        obs = g['obs']
    assert congested > 1000
    assert worst <= 2.0 ** -16, worst                           # a few snapped neighbours add up; never beyond one grid step
    REPORT['default_grid_caltech'] = {'stationarity_amps': worst, 'congested': congested}
    dbg.close()
