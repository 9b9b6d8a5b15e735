"""Optimality certificate of an action projection, written against the reference's PROBLEM statement only
(test infrastructure; numpy / SciPy, nothing from ``oracle/`` and no solver of this repository):

    env.py:178-198   minimise ||x - a||_2   s.t.   0 <= x <= min(1, demands / A_PERS_TO_KWH / 32),
    env.py:473-500                                  |A~ x| * 32 <= magnitudes,   A~ = A * exp(j deg2rad(phase)).

The objective is strictly convex and the feasible set convex with an interior point (x = 0 where every
magnitude is > 0), so a point is THE projection — what cvxpy / MOSEK return to their tolerance — iff it is
feasible and ``a - x`` lies in the normal cone of the feasible set at x (KKT):

    32 (a - x) = sum_c lambda_c grad|A~_c y| (y = 32 x)  +  mu_upper - mu_lower,      lambda, mu >= 0,
    lambda_c > 0 only on rows AT their limit, mu only on coordinates AT their bound.

``certify`` works in amps (y = 32 x, b = 32 a, h = 32 u) on a batch: it finds multipliers by least squares on
the free coordinates (exact when they exist and are unique), falls back to ``scipy.optimize.nnls`` over all
active constraints where that fails, and reports per instance the feasibility, the stationarity residual, the
smallest multiplier and the complementarity gap.  It never looks at how the point was produced.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

TIMESTEP_DURATION = 5.0                                   # env.py:99
ACTION_SCALE_FACTOR = 32.0                                # env.py:100
VOLTAGE = 208.0                                           # env.py:103
A_MINS_TO_KWH = (1 / 60) * (VOLTAGE / 1000)               # env.py:108
A_PERS_TO_KWH = A_MINS_TO_KWH * TIMESTEP_DURATION         # env.py:111


def a_tilde(constraint_matrix: np.ndarray, phase_angles_deg: np.ndarray) -> np.ndarray:
    """env.py:485-486."""
    phase_factor = np.exp(1j * np.deg2rad(phase_angles_deg))
    return constraint_matrix * phase_factor[None, :]


def upper_bound_amps(demands_f32: np.ndarray) -> np.ndarray:
    """32 * min(1, demands / A_PERS_TO_KWH / 32) (env.py:188-189), demands = the float32 observation (env.py:218)."""
    d = np.asarray(demands_f32, dtype=np.float32).astype(np.float64)
    return np.minimum(1.0, d / A_PERS_TO_KWH / ACTION_SCALE_FACTOR) * ACTION_SCALE_FACTOR


@dataclass
class Certificate:
    moved: np.ndarray          # [B] bool: the point differs from clip(b, 0, h) (the projection did something)
    n_active: np.ndarray       # [B] rows at their limit
    row_excess: np.ndarray     # [B] max_c |A~_c y| / magnitude_c - 1 (<= 0: inside)
    box_excess: np.ndarray     # [B] max(-y, y - h) in amps
    stationarity: np.ndarray   # [B] max-norm (amps) of what is left of b - y outside the normal cone
    min_lambda: np.ndarray     # [B] smallest row multiplier (0 if none)
    complementarity: np.ndarray  # [B] max_c lambda_c * (magnitude_c - |A~_c y|)+  (A^2)
    used_nnls: np.ndarray      # [B] bool

    def worst(self) -> dict:
        return {k: float(np.max(getattr(self, k))) for k in ('row_excess', 'box_excess', 'stationarity', 'complementarity')} | \
               {'min_lambda': float(np.min(self.min_lambda)), 'instances': int(len(self.moved)), 'moved': int(self.moved.sum()),
                'with_active_rows': int((self.n_active > 0).sum()), 'nnls': int(self.used_nnls.sum())}


def _residual_outside_cone(r, lower, upper):
    """What of r = b - y - sum lambda grad is NOT absorbed by box multipliers: r itself on free coordinates, its
    negative part where only the upper bound is active, its positive part where only the lower one is."""
    out = np.where(lower & upper, 0.0, r)
    out = np.where(upper & ~lower, np.minimum(out, 0.0), out)
    out = np.where(lower & ~upper, np.maximum(out, 0.0), out)
    return out


def certify(At: np.ndarray, magnitudes: np.ndarray, b: np.ndarray, h: np.ndarray, y: np.ndarray,
            active_rtol: float = 1e-7, box_atol: float = 1e-9, accept: float = 1e-8) -> Certificate:
    """b, h, y: float64 [B, n] in amps (target, upper bound, candidate projection)."""
    from scipy.optimize import nnls
    b = np.atleast_2d(np.asarray(b, np.float64)); h = np.atleast_2d(np.asarray(h, np.float64)); y = np.atleast_2d(np.asarray(y, np.float64))
    B, n = y.shape
    m = len(magnitudes)
    w = y @ At.T                                             # [B, m] complex row currents
    f = np.abs(w)
    row_excess = np.max(f / magnitudes - 1.0, axis=1)
    box_excess = np.maximum(np.max(-y, axis=1), np.max(y - h, axis=1))
    box = np.clip(b, 0.0, h)
    moved = np.any(y != box, axis=1)
    active = f >= magnitudes * (1.0 - active_rtol)            # [B, m]
    lower = y <= box_atol
    upper = y >= h - box_atol
    d = b - y
    stat = np.zeros(B); min_lam = np.zeros(B); comp = np.zeros(B); used = np.zeros(B, bool)
    none = ~active.any(axis=1)
    if none.any():                                            # no row at its limit: the box alone must explain b - y
        stat[none] = np.max(np.abs(_residual_outside_cone(d[none], lower[none], upper[none])), axis=1)
    idx = np.flatnonzero(~none)
    for lo in range(0, len(idx), 4096):
        ii = idx[lo:lo + 4096]
        wa = w[ii]; fa = np.where(active[ii], f[ii], 1.0)
        # grad_y |A~_c y| = Re(conj(w_c) A~_c) / |w_c|
        g = (np.conj(wa)[:, :, None] * At[None, :, :]).real / fa[:, :, None] * active[ii][:, :, None]     # [b, m, n]
        free = ~(lower[ii] | upper[ii])
        gf = g * free[:, None, :]
        K = gf @ gf.transpose(0, 2, 1)
        rhs = gf @ d[ii][:, :, None]
        lam = (np.linalg.pinv(K, hermitian=True) @ rhs)[:, :, 0] * active[ii]
        r = _residual_outside_cone(d[ii] - np.einsum('bm,bmn->bn', lam, g), lower[ii], upper[ii])
        s = np.max(np.abs(r), axis=1)
        ml = np.min(np.where(active[ii], lam, np.inf), axis=1)
        bad = np.flatnonzero((s > accept) | (ml < -accept))
        for k in bad:                                         # degenerate multipliers: non-negative least squares over rows AND box
            cols = [g[k, c] for c in np.flatnonzero(active[ii][k])]
            nrow = len(cols)
            up = np.flatnonzero(upper[ii][k]); dn = np.flatnonzero(lower[ii][k])
            E = np.zeros((n, len(up) + len(dn)))
            E[up, np.arange(len(up))] = 1.0
            E[dn, len(up) + np.arange(len(dn))] = -1.0
            M = np.concatenate([np.array(cols).T, E], axis=1)
            z, _ = nnls(M, d[ii][k], maxiter=20 * M.shape[1])
            rr = d[ii][k] - M @ z
            if np.max(np.abs(rr)) < s[k] or ml[k] < -accept:
                s[k] = np.max(np.abs(rr))
                lam[k] = 0.0
                lam[k, np.flatnonzero(active[ii][k])] = z[:nrow]
                ml[k] = z[:nrow].min() if nrow else 0.0
                used[ii[k]] = True
        stat[ii] = s
        min_lam[ii] = ml
        comp[ii] = np.max(np.maximum(lam, 0.0) * np.maximum(magnitudes - f[ii], 0.0), axis=1)
    return Certificate(moved, active.sum(axis=1), row_excess, box_excess, stat, min_lam, comp, used)


def scipy_projection(At, magnitudes, b, h):
    """The same problem through SciPy's SLSQP from a cold start (amps): an independent general-purpose solver, for a
    handful of instances (it is slow and only ~1e-6 A accurate)."""
    from scipy.optimize import minimize
    n = len(b)
    cons = [{'type': 'ineq', 'fun': (lambda v, c=c: magnitudes[c] ** 2 - np.abs(At[c] @ v) ** 2),
             'jac': (lambda v, c=c: -2.0 * (np.conj(At[c] @ v) * At[c]).real)} for c in range(len(magnitudes))]
    res = minimize(lambda v: 0.5 * np.sum((v - b) ** 2), np.clip(b, 0, h) * 0.5, jac=lambda v: v - b,
                   bounds=list(zip(np.zeros(n), h)), constraints=cons, method='SLSQP', options={'ftol': 1e-16, 'maxiter': 500})
    return res.x
