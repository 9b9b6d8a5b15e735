"""Prototype 2: conic-dual LM-Newton ascent for the action projection (numpy vs SciPy).

Problem in amps (y = 32 x):   min 0.5||y-b||^2  s.t. 0<=y<=h, ||M_c S(y)|| <= r_c
Dual (z_c in R^2 per constraint):
    q(z) = min_{0<=y<=h} 0.5||y-b||^2 + sum_c (z_c . M_c S(y) - r_c ||z_c||)
    inner minimiser closed form: y_i = clip(b_i - nu_g(i), 0, h_i), nu = sum_c M_c' z_c
    grad_c = w_c - r_c z_c/||z_c||  (w_c = M_c S),   z_c = 0 optimal iff ||w_c|| <= r_c
q is concave; ascent with a Levenberg-Marquardt Newton direction and a line search on the sign
of the directional derivative (one station pass per trial point).
"""
from __future__ import annotations

import sys
import numpy as np

sys.path.insert(0, '/root/repo')
sys.path.insert(0, '/root/repo/tools')
from sustaingym_amd.network import caltech_acn, jpl_acn, station_groups  # noqa: E402
from scipy.optimize import minimize  # noqa: E402


def scipy_ref(net, b, h):
    At = net.a_tilde()
    r = net.magnitudes
    cons = [{'type': 'ineq', 'fun': (lambda y, c=c: r[c] ** 2 - np.abs(At[c] @ y) ** 2),
             'jac': (lambda y, c=c: -2 * (np.real(At[c] @ y) * np.real(At[c]) + np.imag(At[c] @ y) * np.imag(At[c])))}
            for c in range(len(r))]
    res = minimize(lambda y: 0.5 * np.sum((y - b) ** 2), np.minimum(b, h) * 0.5, jac=lambda y: y - b,
                   bounds=[(0.0, hi) for hi in h], constraints=cons, method='SLSQP',
                   options={'ftol': 1e-16, 'maxiter': 500})
    return res.x, res



def group_tables(net):
    gid, rep = station_groups(net)
    G = len(rep)
    rad = np.deg2rad(net.phase_angles[rep])
    Mre = net.constraint_matrix[:, rep] * np.cos(rad)[None, :]
    Mim = net.constraint_matrix[:, rep] * np.sin(rad)[None, :]
    M = np.stack([Mre, Mim], axis=1)  # [m, 2, G]
    return gid, G, M


class Solver:
    def __init__(self, net, tables=None):
        self.net = net
        self.gid, self.G, self.M = tables if tables is not None else group_tables(net)
        self.m = len(net.magnitudes)
        self.r = net.magnitudes
        self.passes = 0

    def station_pass(self, z, b, h):
        self.passes += 1
        nu = np.einsum('cag,ca->g', self.M, z)
        v = b - nu[self.gid]
        y = np.clip(v, 0.0, h)
        free = (v > 0.0) & (v <= h) & (h > 0)
        S = np.bincount(self.gid, weights=y, minlength=self.G)
        k = np.bincount(self.gid, weights=free.astype(float), minlength=self.G)
        w = np.einsum('cag,g->ca', self.M, S)
        return y, S, k, w

    def grad(self, z, w):
        """gradient for active rows, and activity info"""
        nz = np.linalg.norm(z, axis=1)
        nw = np.linalg.norm(w, axis=1)
        g = np.zeros_like(z)
        on = nz > 0
        g[on] = w[on] - self.r[on, None] * z[on] / nz[on, None]
        return g, nz, nw

    def project(self, b, h, tol=1e-10, maxit=80, verbose=False):
        m, r = self.m, self.r
        z = np.zeros((m, 2))
        y, S, k, w = self.station_pass(z, b, h)
        nw = np.linalg.norm(w, axis=1)
        if np.all(nw <= r * (1 + tol)):
            return y, z, 0, True
        mu = getattr(self, 'mu0', 1e-3)
        shrink = getattr(self, 'shrink', 0.25)
        for it in range(1, maxit + 1):
            g, nz, nw = self.grad(z, w)
            # activate violated rows with z = 0: tiny multiplier along w
            newly = (nz == 0) & (nw > r * (1 + tol))
            if np.any(newly):
                z = z.copy()
                if getattr(self, 'smart_start', False) and (newly.sum() == 1 or '--scaled' in sys.argv) and not np.any(nz > 0):
                    for c in np.where(newly)[0]:
                        wh = w[c] / nw[c]
                        pr = self.M[c, 0] * wh[0] + self.M[c, 1] * wh[1]
                        curv = float(np.sum(k * pr * pr))
                        lam = (nw[c] - r[c]) / curv if curv > 0 else 1e-6
                        z[c] = max(lam / newly.sum(), 1e-6) * wh
                else:
                    z[newly] = 1e-6 * w[newly] / nw[newly, None]
                y, S, k, w = self.station_pass(z, b, h)
                g, nz, nw = self.grad(z, w)
            A = np.where(nz > 0)[0]
            res_act = np.max(np.linalg.norm(g[A], axis=1) / r[A]) if len(A) else 0.0
            res_inact = np.max((nw / r - 1.0)[nz == 0]) if np.any(nz == 0) else 0.0
            if verbose:
                print(it, 'act', A, 'res', res_act, res_inact, 'mu', mu, '|z|', nz[A])
            if res_act <= tol and res_inact <= tol:
                return y, z, it, True
            # Newton system on active rows
            MA = self.M[A].reshape(2 * len(A), self.G)
            H = (MA * k[None, :]) @ MA.T
            for j, c in enumerate(A):
                zh = z[c] / nz[c]
                H[2 * j:2 * j + 2, 2 * j:2 * j + 2] += (r[c] / nz[c]) * (np.eye(2) - np.outer(zh, zh))
            gA = g[A].reshape(-1)
            scale = max(np.trace(H) / len(gA), 1e-12)
            d = np.linalg.solve(H + mu * scale * np.eye(len(gA)), gA).reshape(len(A), 2)

            def trial(alpha):
                zt = z.copy()
                zt[A] = z[A] + alpha * d
                # a row whose multiplier would cross zero radially is deactivated
                for j, c in enumerate(A):
                    if np.dot(zt[c], z[c]) <= 0.0:
                        zt[c] = 0.0
                st = self.station_pass(zt, b, h)
                gt, nzt, nwt = self.grad(zt, st[3])
                # directional derivative along the actual displacement
                dd = np.sum(gt[A] * (zt[A] - z[A])) / max(alpha, 1e-300)
                # rows deactivated: subgradient; count positive part of violation
                return zt, st, dd

            dd0 = np.sum(g[A] * d)
            alpha = 1.0
            zt, st, dd = trial(alpha)
            if dd > 0.25 * dd0:
                # undershoot (flat region): expand
                best = (zt, st)
                while dd > 0.25 * dd0 and alpha < 1e6:
                    alpha *= 4.0
                    zt2, st2, dd2 = trial(alpha)
                    if dd2 < -0.5 * dd0:
                        break
                    best = (zt2, st2)
                    dd = dd2
                zt, st = best
                mu = max(mu * 0.1, 1e-12)
            else:
                nback = 0
                while dd < -0.5 * dd0 and alpha > 1e-8:
                    alpha *= 0.5
                    nback += 1
                    zt, st, dd = trial(alpha)
                mu = mu * 4.0 if nback > 1 else max(mu * shrink, 1e-12)
            z = zt
            y, S, k, w = st
        return y, z, maxit, False


def main():
    rng = np.random.default_rng(0)
    for net in (caltech_acn(), jpl_acn()):
        sol = Solver(net)
        sol.smart_start = '--smart' in sys.argv
        for a in sys.argv:
            if a.startswith('--mu0='): sol.mu0 = float(a[6:])
            if a.startswith('--shrink='): sol.shrink = float(a[9:])
        print(net.site, 'G =', sol.G)
        n = net.num_stations
        worst = 0.0
        its, passes = [], []
        fails = 0
        for trial in range(400):
            occ = rng.random(n) < rng.choice([0.2, 0.5, 0.9, 1.0])
            h = np.where(occ, np.minimum(32.0, rng.uniform(0, 60, n) / (208 / 12000)), 0.0)
            if trial % 7 == 0:
                h = np.where(occ, rng.uniform(0, 32, n), 0.0)
            mode = trial % 4
            if mode == 0:
                b = rng.uniform(0, 32, n)
            elif mode == 1:
                b = np.full(n, 32.0)
            elif mode == 2:
                b = 32.0 * (rng.random(n) < 0.7)
            else:
                b = rng.uniform(16, 32, n)
            sol.passes = 0
            y, z, it, ok = sol.project(b, h)
            if it > 0:
                its.append(it)
                passes.append(sol.passes)
            if not ok:
                fails += 1
                print('FAIL trial', trial)
                continue
            if it > 0 and trial % 3 == 0:
                yr, res = scipy_ref(net, b, h)
                d = np.max(np.abs(y - yr))
                worst = max(worst, d)
                if d > 1e-4:
                    fo, fr = 0.5 * np.sum((y - b) ** 2), 0.5 * np.sum((yr - b) ** 2)
                    print('trial', trial, 'diff', d, 'obj', fo, fr, res.status, 'viol ours',
                          np.max(np.abs(net.a_tilde() @ y) - net.magnitudes), 'ref',
                          np.max(np.abs(net.a_tilde() @ yr) - net.magnitudes))
        print('slow', len(its), 'fails', fails, 'iters mean/max', np.mean(its), max(its),
              'passes mean/max', np.mean(passes), max(passes), 'worst diff', worst)


if __name__ == '__main__':
    main()
