/*
 * evc_oracle_proj.c — CPU ORACLE (test infrastructure): the action projection.
 *
 * Restates the PROBLEM of env.py:178-221 + magnitude_constraint (env.py:473-500):
 *
 *     x* = argmin ||x - a||_2
 *          s.t. 0 <= x <= min(1, demands / A_PERS_TO_KWH / 32)             (env.py:181,188-193)
 *               |A~ x| * 32 <= magnitudes,  A~ = constraint_matrix * exp(j deg2rad(phase))
 *
 * The reference hands this SOCP to cvxpy + MOSEK (sustaingym/envs/utils.py:6-24); neither is
 * available here, and an interior-point answer is only eps-accurate anyway.
 * This file solves the same problem to a verified KKT tolerance with its own method, written
 * directly on the literal complex rows (no station classes, unlike the HIP kernel):
 *
 *   conic dual   q(z) = min_{0<=y<=h} 1/2||y-b||^2 + sum_c ( z_c . B_c y - r_c ||z_c|| ),
 *   y = 32 x (amps), B_c = [Re A~_c ; Im A~_c] in R^{2 x n}, z_c in R^2.
 *   The inner minimiser is the clip y_i = clip(b_i - (B' z)_i, 0, h_i); q is concave with
 *   gradient w_c - r_c z_c/||z_c|| (w_c = B_c y) and z_c = 0 is optimal iff ||w_c|| <= r_c.
 *   Ascent: Levenberg-Marquardt Newton direction on the active rows + a line search on the
 *   sign of the directional derivative (expansion on flat pieces, bisection on overshoot).
 *
 * The returned point carries a KKT certificate evaluated from primal quantities
 * (kkt_out), which tests assert; tests also cross-check against SciPy SLSQP.
 */
#include "evc_oracle.h"
#include "evc_oracle_priv.h"

#include <math.h>
#include <string.h>

#define NMAX ORC_MAX_STATIONS
#define MMAX ORC_MAX_CONSTRAINTS
#define DMAX (2 * MMAX)


typedef struct {
    int n, m;
    const double* b;
    const double* h;
    double Bre[MMAX][NMAX], Bim[MMAX][NMAX];
    const double* r;
    /* state of the last pass */
    double y[NMAX];
    unsigned char is_free[NMAX];
    double w[MMAX][2];
} pctx;

static void station_pass(pctx* p, double z[MMAX][2]) {
    const int n = p->n, m = p->m;
    for (int i = 0; i < n; i++) {
        double nu = 0.0;
        for (int c = 0; c < m; c++) {
            if (z[c][0] == 0.0 && z[c][1] == 0.0) continue;
            nu += p->Bre[c][i] * z[c][0] + p->Bim[c][i] * z[c][1];
        }
        double v = p->b[i] - nu;
        double y = v;
        if (y > p->h[i]) y = p->h[i];
        if (y < 0.0) y = 0.0;
        p->y[i] = y;
        p->is_free[i] = (v > 0.0 && v <= p->h[i] && p->h[i] > 0.0);
    }
    for (int c = 0; c < m; c++) {
        double re = 0.0, im = 0.0;
        for (int i = 0; i < n; i++) {
            re += p->Bre[c][i] * p->y[i];
            im += p->Bim[c][i] * p->y[i];
        }
        p->w[c][0] = re;
        p->w[c][1] = im;
    }
}

/* dense SPD solve (Cholesky with tiny pivot floor); H is d x d row-major with stride DMAX */
static void spd_solve(double H[DMAX][DMAX], double* rhs, int d) {
    for (int j = 0; j < d; j++) {
        double s = H[j][j];
        for (int k = 0; k < j; k++) s -= H[j][k] * H[j][k];
        if (s < 1e-300) s = 1e-300;
        double l = sqrt(s);
        H[j][j] = l;
        for (int i = j + 1; i < d; i++) {
            double t = H[i][j];
            for (int k = 0; k < j; k++) t -= H[i][k] * H[j][k];
            H[i][j] = t / l;
        }
    }
    for (int i = 0; i < d; i++) {
        double t = rhs[i];
        for (int k = 0; k < i; k++) t -= H[i][k] * rhs[k];
        rhs[i] = t / H[i][i];
    }
    for (int i = d - 1; i >= 0; i--) {
        double t = rhs[i];
        for (int k = i + 1; k < d; k++) t -= H[k][i] * rhs[k];
        rhs[i] = t / H[i][i];
    }
}


static void gradient(const pctx* p, double z[MMAX][2], double g[MMAX][2], double* nz, double* nw) {
    for (int c = 0; c < p->m; c++) {
        nz[c] = hypot(z[c][0], z[c][1]);
        nw[c] = hypot(p->w[c][0], p->w[c][1]);
        if (nz[c] > 0.0) {
            g[c][0] = p->w[c][0] - p->r[c] * z[c][0] / nz[c];
            g[c][1] = p->w[c][1] - p->r[c] * z[c][1] / nz[c];
        } else {
            g[c][0] = g[c][1] = 0.0;
        }
    }
}

/* evaluate z + alpha d (rows crossing zero radially are deactivated); leaves the pass state
 * of the trial point in p and returns the directional derivative per unit alpha */
static double try_step(pctx* p, double z[MMAX][2], const int* act, int na, const double* d,
                       double alpha, double zt[MMAX][2]) {
    memcpy(zt, z, sizeof(double) * MMAX * 2);
    for (int j = 0; j < na; j++) {
        int c = act[j];
        double t0 = z[c][0] + alpha * d[2 * j], t1 = z[c][1] + alpha * d[2 * j + 1];
        if (t0 * z[c][0] + t1 * z[c][1] <= 0.0) t0 = t1 = 0.0;
        zt[c][0] = t0;
        zt[c][1] = t1;
    }
    station_pass(p, zt);
    double g[MMAX][2], nz[MMAX], nw[MMAX];
    gradient(p, zt, g, nz, nw);
    double dd = 0.0;
    for (int j = 0; j < na; j++) {
        int c = act[j];
        dd += g[c][0] * (zt[c][0] - z[c][0]) + g[c][1] * (zt[c][1] - z[c][1]);
    }
    return dd / alpha;
}

/* Safeguard for the Newton ascent above (it can stall where many overlapping rows make the multipliers
 * non-unique): accelerated proximal gradient on the same dual.  q(z) = f(z) - sum_c r_c ||z_c|| with
 * f(z) = min_{0<=y<=h} 1/2||y-b||^2 + z.By concave, grad f = B y(z), Lipschitz with L = lambda_max(B B');
 * the prox of the norm term is a block soft-threshold.  FISTA with gradient restart: globally convergent,
 * linear in practice (50-550 passes on 1 050 random networks / demands, tools/proj_fallback_proto.py), only
 * ever entered when the Newton did not deliver.  L is bounded by Gershgorin on B B'.  Leaves the state of
 * the returned z in p; returns 1 if the KKT residuals of the Newton's convergence test are met. */
static int dual_proximal_gradient(pctx* p, double z[MMAX][2], double tol, double tol_kkt) {
    const int n = p->n, m = p->m;
    double L = 0.0;
    for (int a = 0; a < 2 * m; a++) {
        const double* ra = (a & 1) ? p->Bim[a >> 1] : p->Bre[a >> 1];
        double row = 0.0;
        for (int bb = 0; bb < 2 * m; bb++) {
            const double* rb = (bb & 1) ? p->Bim[bb >> 1] : p->Bre[bb >> 1];
            double dot = 0.0;
            for (int i = 0; i < n; i++) dot += ra[i] * rb[i];
            row += fabs(dot);
        }
        if (row > L) L = row;
    }
    if (!(L > 0.0)) return 0;
    const double t = 1.0 / L;
    double v[MMAX][2], zn[MMAX][2];
    memset(z, 0, sizeof(double) * MMAX * 2);
    memset(v, 0, sizeof(v));
    double theta = 1.0;
    for (int k = 0; k < 200000; k++) {
        station_pass(p, v);
        double restart = 0.0;
        for (int c = 0; c < m; c++) {
            double u0 = v[c][0] + t * p->w[c][0], u1 = v[c][1] + t * p->w[c][1];
            double nu = hypot(u0, u1);
            double shrink = nu > 0.0 ? 1.0 - t * p->r[c] / nu : 0.0;
            if (shrink < 0.0) shrink = 0.0;
            zn[c][0] = u0 * shrink;
            zn[c][1] = u1 * shrink;
            restart += (zn[c][0] - z[c][0]) * (v[c][0] - zn[c][0]) + (zn[c][1] - z[c][1]) * (v[c][1] - zn[c][1]);
        }
        double theta_n = 1.0, beta = 0.0;
        if (!(restart > 0.0)) {
            theta_n = 0.5 * (1.0 + sqrt(1.0 + 4.0 * theta * theta));
            beta = (theta - 1.0) / theta_n;
        }
        for (int c = 0; c < m; c++)
            for (int q = 0; q < 2; q++) {
                v[c][q] = zn[c][q] + beta * (zn[c][q] - z[c][q]);
                z[c][q] = zn[c][q];
            }
        theta = theta_n;
        if (k % 16 == 15) {
            station_pass(p, z);
            double g[MMAX][2], nz[MMAX], nw[MMAX], res_act = 0.0, res_inact = 0.0;
            gradient(p, z, g, nz, nw);
            for (int c = 0; c < m; c++) {
                double rr = nz[c] > 0.0 ? hypot(g[c][0], g[c][1]) / p->r[c] : nw[c] / p->r[c] - 1.0;
                if (nz[c] > 0.0) { if (rr > res_act) res_act = rr; }
                else if (rr > res_inact) res_inact = rr;
            }
            if (res_act <= tol_kkt && res_inact <= tol) return 1;
        }
    }
    station_pass(p, z);
    return 0;
}

int orc_project_action_impl(const orc_net* net, const double* action, const float* demands,
                            double* x_out, double* kkt_out) {
    const int n = net->n, m = net->m;
    const double TOL = 1e-10;      /* feasibility: rows within (1+TOL) r count as satisfied   */
    const double TOL_KKT = 1e-12;  /* dual-gradient residual the Newton ascent is driven to   */
    const double TOL_ACCEPT = 1e-9; /* accepted (status ok) if the iteration budget runs out  */
    /* env.py:108,111 */
    const double A_MINS_TO_KWH = (1.0 / 60.0) * (208.0 / 1000.0);
    const double A_PERS_TO_KWH = A_MINS_TO_KWH * 5.0;
    double b[NMAX], h[NMAX];
    for (int i = 0; i < n; i++) {
        double u = (double)demands[i] / A_PERS_TO_KWH / 32.0; /* env.py:188-189 */
        if (u > 1.0) u = 1.0;
        b[i] = action[i] * 32.0;
        h[i] = u * 32.0;
    }
    static __thread pctx P; /* large; keep off the stack */
    pctx* p = &P;
    p->n = n;
    p->m = m;
    p->b = b;
    p->h = h;
    p->r = net->mag;
    for (int c = 0; c < m; c++)
        for (int i = 0; i < n; i++) {
            p->Bre[c][i] = net->A[c * n + i] * net->cosphi[i]; /* env.py:485-486 */
            p->Bim[c][i] = net->A[c * n + i] * net->sinphi[i];
        }
    double z[MMAX][2];
    memset(z, 0, sizeof(z));
    station_pass(p, z);
    int converged = 0, need = 0;
    for (int c = 0; c < m; c++)
        if (hypot(p->w[c][0], p->w[c][1]) > p->r[c] * (1.0 + TOL)) need = 1;
    if (!need) converged = 1;

    double mu = 1e-3;
    int last_ok = 0;
    for (int it = 0; it < 60 && !converged; it++) {
        double g[MMAX][2], nz[MMAX], nw[MMAX];
        gradient(p, z, g, nz, nw);
        int newly = 0;
        for (int c = 0; c < m; c++)
            if (nz[c] == 0.0 && nw[c] > p->r[c] * (1.0 + TOL)) {
                z[c][0] = 1e-6 * p->w[c][0] / nw[c];
                z[c][1] = 1e-6 * p->w[c][1] / nw[c];
                newly = 1;
            }
        if (newly) {
            station_pass(p, z);
            gradient(p, z, g, nz, nw);
        }
        int act[MMAX], na = 0;
        double res_act = 0.0, res_inact = 0.0;
        for (int c = 0; c < m; c++) {
            if (nz[c] > 0.0) {
                act[na++] = c;
                double rr = hypot(g[c][0], g[c][1]) / p->r[c];
                if (rr > res_act) res_act = rr;
            } else {
                double rr = nw[c] / p->r[c] - 1.0;
                if (rr > res_inact) res_inact = rr;
            }
        }
        if (res_act <= TOL_KKT && res_inact <= TOL) {
            converged = 1;
            break;
        }
        last_ok = (res_act <= TOL_ACCEPT && res_inact <= TOL);
        /* H = B_A diag(free) B_A' + tangential curvature + LM */
        static __thread double H[DMAX][DMAX];
        double rhs[DMAX];
        const int d = 2 * na;
        for (int a = 0; a < d; a++)
            for (int bb = 0; bb < d; bb++) H[a][bb] = 0.0;
        for (int i = 0; i < n; i++) {
            if (!p->is_free[i]) continue;
            double col[DMAX];
            for (int j = 0; j < na; j++) {
                col[2 * j] = p->Bre[act[j]][i];
                col[2 * j + 1] = p->Bim[act[j]][i];
            }
            for (int a = 0; a < d; a++) {
                if (col[a] == 0.0) continue;
                for (int bb = 0; bb < d; bb++) H[a][bb] += col[a] * col[bb];
            }
        }
        for (int j = 0; j < na; j++) {
            int c = act[j];
            double zh0 = z[c][0] / nz[c], zh1 = z[c][1] / nz[c];
            double s = p->r[c] / nz[c];
            H[2 * j][2 * j] += s * (1.0 - zh0 * zh0);
            H[2 * j][2 * j + 1] += s * (-zh0 * zh1);
            H[2 * j + 1][2 * j] += s * (-zh0 * zh1);
            H[2 * j + 1][2 * j + 1] += s * (1.0 - zh1 * zh1);
            rhs[2 * j] = g[c][0];
            rhs[2 * j + 1] = g[c][1];
        }
        double tr = 0.0;
        for (int a = 0; a < d; a++) tr += H[a][a];
        double scale = tr / d;
        if (scale < 1e-12) scale = 1e-12;
        for (int a = 0; a < d; a++) H[a][a] += mu * scale;
        double dir[DMAX];
        memcpy(dir, rhs, sizeof(double) * d);
        spd_solve(H, dir, d);
        double dd0 = 0.0;
        for (int a = 0; a < d; a++) dd0 += rhs[a] * dir[a];

        double zt[MMAX][2], zbest[MMAX][2];
        double alpha = 1.0;
        double dd = try_step(p, z, act, na, dir, alpha, zt);
        if (dd > 0.25 * dd0) { /* undershoot: expand */
            memcpy(zbest, zt, sizeof(zt));
            while (dd > 0.25 * dd0 && alpha < 1e6) {
                alpha *= 4.0;
                double dd2 = try_step(p, z, act, na, dir, alpha, zt);
                if (dd2 < -0.5 * dd0) break;
                memcpy(zbest, zt, sizeof(zt));
                dd = dd2;
            }
            memcpy(z, zbest, sizeof(zbest));
            mu = mu * 0.1;
            if (mu < 1e-12) mu = 1e-12;
        } else {
            int nback = 0;
            while (dd < -0.5 * dd0 && alpha > 1e-8) {
                alpha *= 0.5;
                nback++;
                dd = try_step(p, z, act, na, dir, alpha, zt);
            }
            memcpy(z, zt, sizeof(zt));
            if (nback > 1) mu *= 4.0;
            else {
                mu *= 0.25;
                if (mu < 1e-12) mu = 1e-12;
            }
        }
        station_pass(p, z); /* state of the accepted point */
    }

    if (!converged && last_ok) converged = 1;
    if (!converged) converged = dual_proximal_gradient(p, z, TOL, TOL_KKT);
    /* Tie snap: values the solver moved are snapped to a 2^-16 A grid before the reference's
     * rounding rule (env.py:373-378) sees them.  Exact optima that sit on a rounding boundary
     * (e.g. an 80 A pod shared by 4 EVs -> 20 A -> rint(2.5)) are thereby rounded the same way by
     * every solver that is within ~1e-6 A of the optimum; the reference's interior-point answer
     * is itself only that accurate there.  The grid is offset by sqrt(2)-1 grid
     * steps so that its own midpoints are never dyadic / small-denominator rationals, which exact
     * optima (e.g. (sum b - 80)/4) frequently are. */
    for (int i = 0; i < n; i++) {
        double y0 = b[i] < h[i] ? b[i] : h[i];
        double y = p->y[i];
        if (y != y0) {
            y = (rint(y * 65536.0 - 0.41421356237309515) + 0.41421356237309515) / 65536.0;
            if (y < 0.0) y = 0.0;
            if (y > h[i]) y = h[i];
        }
        x_out[i] = y / 32.0;
    }

    /* ---- KKT certificate from primal quantities and multiplier magnitudes only ---- */
    if (kkt_out) {
        double lam[MMAX], nw[MMAX];
        double infeas = 0.0, cslack = 0.0, minlam = 0.0, align = 0.0;
        for (int c = 0; c < m; c++) {
            lam[c] = hypot(z[c][0], z[c][1]);
            nw[c] = hypot(p->w[c][0], p->w[c][1]);
            double v = (nw[c] - p->r[c]) / p->r[c];
            if (v > infeas) infeas = v;
            double cs = lam[c] * fabs(p->r[c] - nw[c]) / p->r[c];
            if (cs > cslack) cslack = cs;
            if (lam[c] > 0.0 && nw[c] > 0.0) {
                double a0 = z[c][0] / lam[c] - p->w[c][0] / nw[c];
                double a1 = z[c][1] / lam[c] - p->w[c][1] / nw[c];
                double al = hypot(a0, a1);
                if (al > align) align = al;
            }
        }
        double stat = 0.0;
        for (int i = 0; i < n; i++) {
            double grad = 0.0; /* sum_c lam_c d|A~_c y|/dy_i */
            for (int c = 0; c < m; c++) {
                if (lam[c] == 0.0 || nw[c] == 0.0) continue;
                grad += lam[c] * (p->Bre[c][i] * p->w[c][0] + p->Bim[c][i] * p->w[c][1]) / nw[c];
            }
            double v = b[i] - grad;
            if (v > h[i]) v = h[i];
            if (v < 0.0) v = 0.0;
            double e = fabs(v - p->y[i]);
            if (e > stat) stat = e;
        }
        kkt_out[0] = stat;   /* amps */
        kkt_out[1] = infeas; /* relative */
        kkt_out[2] = cslack;
        kkt_out[3] = align > minlam ? align : minlam;
    }
    return converged ? 0 : 1;
}
