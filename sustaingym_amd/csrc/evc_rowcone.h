// evc_rowcone.h — the action projection (env.py:178-221) of FOUR environments at once, one per 16-lane DPP row, for the
// environments whose box-clipped action leaves one or two constraint rows violated (a class cap counts as a row: its own
// simple row of the matrix).  Same mathematics as wave_cone (evc_solver.h: conic dual, Levenberg-Marquardt Newton on the
// active rows, full steps, exact KKT test) in the geometry of the quad kernels: lane q of a row owns stations q, q + 16,
// q + 32, q + 48 of its environment, every sum of an iteration is a per-lane sum over those four slots followed by ONE
// 16-lane butterfly (row_allreduce_f64: 4 DPP steps instead of the wave ladder's 6 + readlane, and four environments
// share each instruction).  Used by the fused rollout kernel (evc_rollout.h), where under a greedy policy nearly every
// environment of a congested network needs such a solve in every period: the wave-per-environment solver behind
// rollout_solve_rows took the rows of a wavefront one after the other (JPL GMM days under greedy: 6e7 env-steps/s).
//
// A row that this file cannot settle — a third row turns up, a multiplier wants to leave the active set, the iteration
// budget — is reported back and goes through the general path (solve_projection) as before: nothing here decides
// feasibility on its own, every accepted point has been verified against ALL rows and caps in float64.
#pragma once

#include "evc_quad.h"

namespace evc {

__device__ __forceinline__ double row_allreduce_max_f64(double v) {
    v = fmax(v, dpp_f64<0x121, 0xf, false>(v));
    v = fmax(v, dpp_f64<0x122, 0xf, false>(v));
    v = fmax(v, dpp_f64<0x124, 0xf, false>(v));
    v = fmax(v, dpp_f64<0x128, 0xf, false>(v));
    return v;
}

// Exact float64 rows of the station-shaped schedule y of MY row's environment (lane q < m evaluates constraint row q):
// viol = the violated rows (bit c), cap_viol = the classes above their cap, worst = the row furthest above its limit
// (-1 if none).  All three are uniform over the 16 lanes of a row.
struct RowExact { unsigned viol; unsigned cap_viol; int worst; };
__device__ __forceinline__ RowExact quad_exact_rows_worst(int G, const double* class_cap, const LdsNet& net, unsigned q, unsigned m,
                                                         unsigned row, const int (&st_gid)[kSlots], const double (&y)[kSlots],
                                                         double tol = Consts::PROJ_TOL) {
    double re = 0.0, im = 0.0;
    RowExact out{0u, 0u, -1};
    for (int g0 = 0; g0 < G; g0 += 4) {
        double S[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            double part = 0.0;
#pragma unroll
            for (int j = 0; j < kSlots; j++) part += (st_gid[j] == g0 + u) ? y[j] : 0.0;
            S[u] = row_allreduce_f64(part);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int g = g0 + u;
            if (g < G) {
                if (q < m) { re += net.Mre[g][q] * S[u]; im += net.Mim[g][q] * S[u]; }
                if (S[u] > class_cap[g] * (1.0 + tol)) out.cap_viol |= 1u << g;
            }
        }
    }
    double ratio = 0.0;
    bool viol = false;
    if (q < m) {
        const double mg = net.mag[q], lim = mg * (1.0 + tol), m2 = re * re + im * im;
        viol = m2 > lim * lim;
        ratio = viol ? m2 / (mg * mg) : 0.0;          // squared ratios order the rows like the ratios do
    }
    out.viol = (unsigned)(__ballot(viol) >> (row * 16u)) & 0xffffu;
    if (__ballot(viol) != 0ull) {
        const double best = row_allreduce_max_f64(ratio);
        const unsigned at = (unsigned)(__ballot(viol && ratio == best) >> (row * 16u)) & 0xffffu;
        out.worst = at ? (int)__builtin_ctz(at) : -1;
    }
    return out;
}

// The conic-dual Newton on R given rows for the rows with `on` (every quantity row-uniform; rows[] and z[] differ between
// the rows of the wavefront).  Returns, per row, whether it converged (yout = the optimum given THESE active rows; the
// caller verifies the other rows).  Mirrors wave_cone<R> statement by statement; see there for the algorithm.
template <int R>
__device__ __forceinline__ bool quad_cone(const LdsNet& net, unsigned row, bool on, const int (&st_gid)[kSlots],
                                          const double (&b)[kSlots], const double (&h)[kSlots], const int (&rows)[R],
                                          double (&z)[2 * R], double (&yout)[kSlots], bool warm = false) {
    // warm (row-uniform): every multiplier of this row's environment is given (last period's optimum): no first-order start
    constexpr int D = 2 * R;
    double cf[D][kSlots], rmag[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int c = rows[r] >= 0 ? rows[r] : 0;
#pragma unroll
        for (int j = 0; j < kSlots; j++) {
            cf[2 * r][j] = st_gid[j] >= 0 ? net.Mre[st_gid[j]][c] : 0.0;
            cf[2 * r + 1][j] = st_gid[j] >= 0 ? net.Mim[st_gid[j]][c] : 0.0;
        }
        rmag[r] = net.mag[c];
    }
    double mu = 1e-3;
    bool run = on, ok = false;
    bool was_free[kSlots];
#pragma unroll
    for (int j = 0; j < kSlots; j++) was_free[j] = false;
    bool have_K = false;
    double K[D][D];
    for (int it = 0; it < (R == 1 ? 14 : 24) && __ballot(run) != 0ull; it++) {
        double y[kSlots];
        bool fr[kSlots], moved = false;
#pragma unroll
        for (int j = 0; j < kSlots; j++) {
            double nu = 0.0;
#pragma unroll
            for (int a = 0; a < D; a++) nu += cf[a][j] * z[a];
            const double v = b[j] - nu;
            y[j] = st_gid[j] >= 0 ? fmin(fmax(v, 0.0), h[j]) : 0.0;
            fr[j] = st_gid[j] >= 0 && (v > 0.0) && (v <= h[j]) && (h[j] > 0.0);
            moved = moved || fr[j] != was_free[j];
            was_free[j] = fr[j];
        }
        double w[D];
#pragma unroll
        for (int a = 0; a < D; a++) {
            double part = 0.0;
#pragma unroll
            for (int j = 0; j < kSlots; j++) part += cf[a][j] * y[j];
            w[a] = row_allreduce_f64(part);
        }
        // K depends on the multipliers only through the set of free stations: summed again when that set changed in any
        // running row (rows whose set did not change get the same values again)
        if (!have_K || __ballot(run && moved) != 0ull) {
#pragma unroll
            for (int a = 0; a < D; a++)
#pragma unroll
                for (int e = a; e < D; e++) {
                    double part = 0.0;
#pragma unroll
                    for (int j = 0; j < kSlots; j++) part += fr[j] ? cf[a][j] * cf[e][j] : 0.0;
                    K[a][e] = row_allreduce_f64(part);
                }
            have_K = true;
        }
        const bool starting = it == 0 && !warm;          // this row only places its new multiplier in this pass
        if (it == 0) {
            // the row without a multiplier: alone, the first-order size along its w; beside an active row, tiny
            const int a0 = D - 2;
            const double nw = sqrt(w[a0] * w[a0] + w[a0 + 1] * w[a0 + 1]);
            if (starting && !(nw > rmag[R - 1])) run = false;           // (not violated after all: the general path decides)
            const double wh0 = w[a0] / nw, wh1 = w[a0 + 1] / nw;
            double lam = 1e-6;
            if (R == 1) {
                const double curv = wh0 * (K[0][0] * wh0 + K[0][1] * wh1) + wh1 * (K[0][1] * wh0 + K[1][1] * wh1);
                if (curv > 0.0) lam = fmax((nw - rmag[0]) / curv, 1e-6);
            }
            if (run && starting) { z[a0] = lam * wh0; z[a0 + 1] = lam * wh1; }
            if (__ballot(run && warm) == 0ull) continue;                // no warm row in this wavefront: nothing else to do in pass 0
        }
        double zh[D], g[D], rn[R];
        bool conv = true;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const double inz = newton_rsqrt(z[2 * r] * z[2 * r] + z[2 * r + 1] * z[2 * r + 1]);
            zh[2 * r] = z[2 * r] * inz;
            zh[2 * r + 1] = z[2 * r + 1] * inz;
            g[2 * r] = w[2 * r] - rmag[r] * zh[2 * r];
            g[2 * r + 1] = w[2 * r + 1] - rmag[r] * zh[2 * r + 1];
            rn[r] = rmag[r] * inz;
            const double lim = Consts::PROJ_TOL_KKT * rmag[r];
            conv = conv && g[2 * r] * g[2 * r] + g[2 * r + 1] * g[2 * r + 1] <= lim * lim;
        }
        if (run && conv && !starting) {
#pragma unroll
            for (int j = 0; j < kSlots; j++) yout[j] = y[j];
            ok = true;
            run = false;
        }
        double B[D][D];
#pragma unroll
        for (int a = 0; a < D; a++)
#pragma unroll
            for (int e = 0; e < D; e++) B[a][e] = a <= e ? K[a][e] : K[e][a];
        double tr = 0.0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int a = 2 * r;
            B[a][a] += rn[r] * (1.0 - zh[a] * zh[a]);
            B[a + 1][a + 1] += rn[r] * (1.0 - zh[a + 1] * zh[a + 1]);
            B[a][a + 1] -= rn[r] * zh[a] * zh[a + 1];
            B[a + 1][a] = B[a][a + 1];
            tr += B[a][a] + B[a + 1][a + 1];
        }
        double scale = tr / (double)D;
        scale = scale < 1e-12 ? 1e-12 : scale;
#pragma unroll
        for (int a = 0; a < D; a++) B[a][a] += mu * scale;
        bool good = true;                          // Gauss elimination without pivoting (SPD + shift)
        double inv[D];
#pragma unroll
        for (int a = 0; a < D; a++) {
            good = good && B[a][a] > 0.0;
            inv[a] = newton_rcp(B[a][a]);
#pragma unroll
            for (int e = a + 1; e < D; e++) {
                const double f = B[e][a] * inv[a];
#pragma unroll
                for (int u = a + 1; u < D; u++) B[e][u] -= f * B[a][u];
                g[e] -= f * g[a];
            }
        }
        double d[D];
#pragma unroll
        for (int a = D - 1; a >= 0; a--) {
            double t = g[a];
#pragma unroll
            for (int u = a + 1; u < D; u++) t -= B[a][u] * d[u];
            d[a] = t * inv[a];
        }
        if (!good && !starting) run = false;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const double t0 = z[2 * r] + d[2 * r], t1 = z[2 * r + 1] + d[2 * r + 1];
            if (!starting && !(t0 * z[2 * r] + t1 * z[2 * r + 1] > 0.0)) run = false;     // the row would leave the active set
        }
        if (run && !starting) {
#pragma unroll
            for (int a = 0; a < D; a++) z[a] += d[a];
        }
        if (!starting) mu = fmax(mu * 0.25, 1e-12);
    }
    return ok;
}

// Water-filling of class g on (b, h) — quad_waterfill's iteration without its tie snap (the caller snaps once, at the end).
__device__ __forceinline__ void quad_waterfill_bh(bool on, int g, const int (&st_gid)[kSlots], const double (&b)[kSlots],
                                                  const double (&h)[kSlots], double cap, double (&y)[kSlots]) {
    bool in_g[kSlots];
#pragma unroll
    for (int j = 0; j < kSlots; j++) in_g[j] = st_gid[j] == g;
    double nu = 0.0, lo = 0.0, hi = 64.0;
    bool run = on;
    for (int it = 0; it < 80 && __ballot(run) != 0ull; it++) {
        double part = 0.0;
        unsigned nfree = 0u;
#pragma unroll
        for (int j = 0; j < kSlots; j++) {
            const double v = b[j] - nu;
            part += in_g[j] ? fmin(fmax(v, 0.0), h[j]) : 0.0;
            nfree += (in_g[j] && v > 0.0 && v <= h[j] && h[j] > 0.0) ? 1u : 0u;
        }
        const double f = row_allreduce_f64(part) - cap;
        const unsigned kfree = row_allreduce_u32(nfree);
        double bp = 0.0;
        if (__builtin_expect(__ballot(run && kfree == 0u && f > 0.0) != 0ull, 0)) {
            double nextbp = 1e300;
#pragma unroll
            for (int j = 0; j < kSlots; j++) nextbp = fmin(nextbp, (in_g[j] && b[j] - nu > h[j]) ? b[j] - h[j] : 1e300);
            bp = row_allreduce_min_f64(nextbp);
        }
        if (run) {
            if (fabs(f) <= 1e-13 * cap) {
                run = false;
            } else {
                if (f > 0.0) lo = nu; else hi = nu;
                double nxt = (kfree > 0u) ? nu + f / (double)kfree : (f > 0.0 ? bp : 0.5 * (lo + hi));
                if (!(nxt > lo && nxt < hi)) nxt = 0.5 * (lo + hi);
                nu = nxt;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kSlots; j++)
        if (on && in_g[j]) y[j] = fmin(fmax(b[j] - nu, 0.0), h[j]);
}

// ONE cone row beside class caps with the caps eliminated — wave_cone_capped (evc_solver.h) in row geometry: the rows with `on`
// solve their own constraint row c (row-uniform, different from row to row) by a 2 x 2 Newton on that row's multiplier; a class
// above its cap contributes exactly its cap to w and nothing to the Jacobian; the inner fillings of the classes left at their cap
// run once, at the end.  yw = the schedule after the caps' filling (kept by classes the row does not load).  Returns per row
// whether it converged; the caller verifies every row and cap of the result.
__device__ __forceinline__ bool quad_cone_capped(int G, const double* class_cap, const LdsNet& net, bool on, int c,
                                                 const int (&st_gid)[kSlots], const double (&b)[kSlots], const double (&h)[kSlots],
                                                 const double (&yw)[kSlots], double (&yout)[kSlots]) {
    const int cc = c >= 0 ? c : 0;
    double cf0[kSlots], cf1[kSlots];
    bool in_row[kSlots];
#pragma unroll
    for (int j = 0; j < kSlots; j++) {
        cf0[j] = st_gid[j] >= 0 ? net.Mre[st_gid[j]][cc] : 0.0;
        cf1[j] = st_gid[j] >= 0 ? net.Mim[st_gid[j]][cc] : 0.0;
        in_row[j] = cf0[j] != 0.0 || cf1[j] != 0.0;
    }
    const double rmag = net.mag[cc];
    double z0 = 0.0, z1 = 0.0, mu = 1e-3;
    bool run = on, ok = false;
    unsigned capped = 0u;
    double v[kSlots], y[kSlots];
    for (int it = 0; it < 20 && __ballot(run) != 0ull; it++) {
        bool fr[kSlots];
#pragma unroll
        for (int j = 0; j < kSlots; j++) {
            v[j] = b[j] - (cf0[j] * z0 + cf1[j] * z1);
            y[j] = in_row[j] ? fmin(fmax(v[j], 0.0), h[j]) : 0.0;
            fr[j] = in_row[j] && v[j] > 0.0 && v[j] <= h[j] && h[j] > 0.0;
        }
        double w0 = 0.0, w1 = 0.0, K00 = 0.0, K01 = 0.0, K11 = 0.0;
        // this pass's capped classes, kept only by rows that are still iterating: a row that converged in an earlier pass keeps the
        // set of ITS last pass (the loop below visits only the classes some running row loads, so for a finished row the pass is
        // incomplete; resetting its set made the final filling skip classes and sent the row to the general path — ADVICE r5)
        unsigned cap_it = 0u;
        for (int g = 0; g < G; g++) {
            const double m0 = net.Mre[g][cc], m1 = net.Mim[g][cc];              // (row-uniform)
            const bool loads = m0 != 0.0 || m1 != 0.0;
            if (__ballot(run && loads) == 0ull) continue;
            double part = 0.0;
            unsigned nf = 0u;
#pragma unroll
            for (int j = 0; j < kSlots; j++) {
                part += st_gid[j] == g ? y[j] : 0.0;
                nf += (st_gid[j] == g && fr[j]) ? 1u : 0u;
            }
            const double Wg = row_allreduce_f64(part);
            const double kg = (double)row_allreduce_u32(nf);
            if (loads) {
                if (Wg > class_cap[g]) {
                    cap_it |= 1u << g;
                    w0 += m0 * class_cap[g]; w1 += m1 * class_cap[g];
                } else {
                    w0 += m0 * Wg; w1 += m1 * Wg;
                    K00 += kg * m0 * m0; K01 += kg * m0 * m1; K11 += kg * m1 * m1;
                }
            }
        }
        if (run) capped = cap_it;
        if (it == 0) {                                   // first-order size along w
            const double nw = sqrt(w0 * w0 + w1 * w1);
            if (!(nw > rmag)) run = false;
            const double wh0 = w0 / nw, wh1 = w1 / nw;
            const double curv = wh0 * (K00 * wh0 + K01 * wh1) + wh1 * (K01 * wh0 + K11 * wh1);
            const double lam = curv > 0.0 ? fmax((nw - rmag) / curv, 1e-6) : 1e-6;
            if (run) { z0 = lam * wh0; z1 = lam * wh1; }
            continue;
        }
        const double inz = newton_rsqrt(z0 * z0 + z1 * z1);
        const double zh0 = z0 * inz, zh1 = z1 * inz;
        const double g0 = w0 - rmag * zh0, g1 = w1 - rmag * zh1;
        const double lim = Consts::PROJ_TOL_KKT * rmag;
        if (run && g0 * g0 + g1 * g1 <= lim * lim) { ok = true; run = false; }
        const double rn = rmag * inz;
        double B00 = K00 + rn * (1.0 - zh0 * zh0), B11 = K11 + rn * (1.0 - zh1 * zh1);
        const double B01 = K01 - rn * zh0 * zh1;
        double scale = 0.5 * (B00 + B11);
        scale = scale < 1e-12 ? 1e-12 : scale;
        B00 += mu * scale; B11 += mu * scale;
        const double det = B00 * B11 - B01 * B01;
        if (!(B00 > 0.0) || !(det > 0.0)) run = false;
        const double idet = newton_rcp(det);
        const double d0 = (B11 * g0 - B01 * g1) * idet, d1 = (B00 * g1 - B01 * g0) * idet;
        const double t0 = z0 + d0, t1 = z1 + d1;
        if (!(t0 * z0 + t1 * z1 > 0.0)) run = false;    // the row would leave the active set
        if (run) { z0 = t0; z1 = t1; }
        mu = fmax(mu * 0.25, 1e-12);
    }
    // (a row that converged stopped updating z and `capped`: they are those of its last pass; v and y are recomputed from z)
    if (__ballot(ok) != 0ull) {
#pragma unroll
        for (int j = 0; j < kSlots; j++) {
            v[j] = b[j] - (cf0[j] * z0 + cf1[j] * z1);
            yout[j] = in_row[j] ? fmin(fmax(v[j], 0.0), h[j]) : yw[j];
        }
        for (int g = 0; g < G; g++) {
            const bool do_g = ok && ((capped >> g) & 1u);
            if (__ballot(do_g) != 0ull) quad_waterfill_bh(do_g, g, st_gid, v, h, class_cap[g], yout);
        }
    }
    return ok;
}

// solve_projection's relaxation sequence for the rows with `on`, in row geometry: exact test of the box clip -> class caps
// by water-filling (exact if every row holds afterwards) -> one violated cap's row + the worst remaining row at once, or the
// chain one cone row -> two.  Returns (row-uniform) whether the row is
// settled; y then holds the projection (values the solver moved are tie-snapped like solve_projection's).
// zw / warm_ok (round 4, greedy policy): zw = this row's [m][2] multipliers of one period ago (LDS), warm_ok (row-uniform) says they
// are valid.  A row with one or two rows active then goes straight to the Newton on those rows from those multipliers — no
// filling, no chain; a row the warm solve does not settle is left to the general path.  Whatever settles a row with at most
// two active rows (a filled cap counts: its multiplier is nu cf / |cf|^2 on its simple row) leaves its multipliers in zw and
// returns stored = true; zw may be null.
__device__ __forceinline__ bool quad_project(int G, const double* class_cap, unsigned simple_rows, unsigned long long* tie_counters, int tie_log2, const LdsNet& net,
                                             unsigned q, unsigned m, unsigned row, bool on, const int (&st_gid)[kSlots],
                                             const bool (&is_cc)[kSlots], const double (&b)[kSlots], const double (&h)[kSlots],
                                             double (&y)[kSlots], double (*zw)[2] = nullptr, bool warm_ok = false, bool* stored = nullptr,
                                             const double* ywin = nullptr, unsigned capv_in = 0u, double tol_in = Consts::PROJ_TOL) {
    // ywin / capv_in (round 5, row-uniform capv_in): the caller has ALREADY clipped to the box and filled the classes in capv_in
    // (the period's body does, on the entries, before it decides that a row needs this call): ywin = that schedule (tie-snapped,
    // hence tol_in = Params::snap_tol for its rows), and the exact rows at the box clip + the fillings are not repeated here.
    double y0[kSlots];
#pragma unroll
    for (int j = 0; j < kSlots; j++) { y0[j] = fmin(b[j], h[j]); y[j] = y0[j]; }
    // what the row solved with, for the next period: up to two rows and their multipliers (-1: none)
    int keep_row[2] = {-1, -1};
    double keep_z[4] = {0.0, 0.0, 0.0, 0.0};
    bool keep = false;
    // warm rows: the rows that carried a multiplier one period ago
    int wrow[2] = {-1, -1};
    double wz[4] = {0.0, 0.0, 0.0, 0.0};
    bool warm1 = false, warm2 = false;
    if (zw != nullptr && __ballot(on && warm_ok) != 0ull) {
        const bool nzq = q < m && (zw[q < m ? q : 0][0] != 0.0 || zw[q < m ? q : 0][1] != 0.0);
        const unsigned act = (unsigned)(__ballot(nzq) >> (row * 16u)) & 0xffffu;
        const int na = __popc(act);
        if (on && warm_ok && na >= 1 && na <= 2) {
            wrow[0] = (int)__builtin_ctz(act);
            wz[0] = zw[wrow[0]][0]; wz[1] = zw[wrow[0]][1];
            if (na == 2) {
                wrow[1] = (int)__builtin_ctz(act & (act - 1u));
                wz[2] = zw[wrow[1]][0]; wz[3] = zw[wrow[1]][1];
                warm2 = true;
            } else {
                warm1 = true;
            }
        }
    }
    const bool warm_row = warm1 || warm2;
    const bool handed = ywin != nullptr && on && !warm_row && capv_in != 0u;
    RowExact e0{0u, 0u, -1};
    if (__ballot(on && !warm_row && !handed) != 0ull) e0 = quad_exact_rows_worst(G, class_cap, net, q, m, row, st_gid, y0);
    if (handed) { e0.viol = 1u; e0.cap_viol = capv_in; e0.worst = -1; }      // (which row: known after the rows of ywin below)
    bool settled = on && !warm_row && e0.viol == 0u && e0.cap_viol == 0u;
    bool open = on && !warm_row && !settled;
    bool d2 = false;                               // a pair of rows to solve together, from the filling or from the chain
    int pair[2] = {-1, -1};
    double zp[2] = {0.0, 0.0};
    // caps first, also beside violated multi-class rows (relaxation argument, evc_solver.h)
    const bool fill = open && e0.cap_viol != 0u;
    if (__ballot(fill) != 0ull) {
        double yw[kSlots];
#pragma unroll
        for (int j = 0; j < kSlots; j++) yw[j] = handed ? ywin[j] : y0[j];
        for (int g = 0; g < G; g++) {
            const bool do_g = fill && !handed && ((e0.cap_viol >> g) & 1u);
            if (__ballot(do_g) != 0ull) quad_waterfill_bh(do_g, g, st_gid, b, h, class_cap[g], yw);
        }
        const RowExact ew = quad_exact_rows_worst(G, class_cap, net, q, m, row, st_gid, yw, handed ? tol_in : Consts::PROJ_TOL);
        if (handed) e0.worst = ew.worst;               // the chain below starts from the worst row it knows
        const bool okw = fill && ew.viol == 0u && ew.cap_viol == 0u;
        if (okw) {
#pragma unroll
            for (int j = 0; j < kSlots; j++) y[j] = yw[j];
        }
        settled = settled || okw;
        open = open && !okw;
#ifndef EVC_NO_ROW_CAPPED_CONE
        // Round 5: two (or more) caps filled and a row still violated — three rows, which this geometry used to hand to the general
        // path (a wavefront per environment, one after the other).  The worst row with the caps eliminated (quad_cone_capped).
        const bool capcone = fill && !okw && ew.worst >= 0 && __popc(e0.cap_viol) >= 2;
        if (__ballot(capcone) != 0ull) {
            double yc[kSlots];
#pragma unroll
            for (int j = 0; j < kSlots; j++) yc[j] = 0.0;
            const bool cc = quad_cone_capped(G, class_cap, net, capcone, ew.worst, st_gid, b, h, yw, yc);
            const RowExact ec = quad_exact_rows_worst(G, class_cap, net, q, m, row, st_gid, yc);
            const bool okc = cc && ec.viol == 0u && ec.cap_viol == 0u;
            if (okc) {
#pragma unroll
                for (int j = 0; j < kSlots; j++) y[j] = yc[j];
            }
            settled = settled || okc;
            open = open && !okc;
        }
#endif
        // The caps' own rows and multipliers, from the filling (a class shifted by nu has z_c = nu cf / |cf|^2 on its simple
        // row c): for a row the filling settled they are what the next period starts from; where a row is still violated, the
        // caps' rows and the worst remaining row are solved together (solve_projection's (b2)).
        const bool direct = fill && !okw && __popc(e0.cap_viol) <= 2 && ew.worst >= 0;
        const bool capinfo = (direct || (okw && zw != nullptr)) && __popc(e0.cap_viol) <= 2;
        if (__ballot(capinfo) != 0ull) {
            int crow[2] = {-1, -1};
            double zc[4] = {0.0, 0.0, 0.0, 0.0};
            int nc = 0;
            bool usable = capinfo;
            for (int g = 0; g < G; g++) {
                const bool mine = capinfo && ((e0.cap_viol >> g) & 1u);
                if (__ballot(mine) == 0ull) continue;
                const bool is_row = q < m && ((simple_rows >> q) & 1u) && (net.Mre[g][q] != 0.0 || net.Mim[g][q] != 0.0);
                const unsigned rb = (unsigned)(__ballot(is_row) >> (row * 16u)) & 0xffffu;
                double cand = -1.0;
#pragma unroll
                for (int j = 0; j < kSlots; j++)
                    if (st_gid[j] == g && yw[j] > 0.0 && yw[j] < h[j] && yw[j] < b[j]) cand = fmax(cand, b[j] - yw[j]);
                const double nu = row_allreduce_max_f64(cand);
                if (mine) {
                    if (rb == 0u || !(nu > 0.0) || nc >= 2) {
                        usable = false;
                    } else {
                        const int c = (int)__builtin_ctz(rb);
                        const double cre = net.Mre[g][c], cim = net.Mim[g][c];
                        const double sc = nu / (cre * cre + cim * cim);
                        if (nc == 0) { crow[0] = c; zc[0] = sc * cre; zc[1] = sc * cim; }
                        else { crow[1] = c; zc[2] = sc * cre; zc[3] = sc * cim; }
                        nc++;
                    }
                }
            }
            if (okw && usable && nc >= 1) {
                keep = true;
                keep_row[0] = crow[0]; keep_row[1] = crow[1];
#pragma unroll
                for (int a = 0; a < 4; a++) keep_z[a] = zc[a];
            }
            usable = usable && direct && nc >= 1 && ew.worst != crow[0] && ew.worst != crow[1];
            // (three rows — both pods beside a feeder row — in this geometry cost every copy of the period's body its registers:
            // scratch 444 -> 1584 B, Caltech GMM greedy 57 -> 78 us per period, measured; those rows go to the general path)
            d2 = usable && nc == 1;
            if (d2) { pair[0] = crow[0]; pair[1] = ew.worst; zp[0] = zc[0]; zp[1] = zc[1]; }
            open = open && !d2;                    // (a direct attempt that fails goes to the general path, not through the chain)
        }
    }
    if (settled && !keep && e0.cap_viol == 0u) keep = true;       // nothing binds: next period starts from zero multipliers
    bool go2 = false;
    const bool on1 = (open && e0.worst >= 0) || warm1;
    if (__ballot(on1) != 0ull) {
        const int r1[1] = {warm1 ? wrow[0] : e0.worst};
        double z1[2] = {warm1 ? wz[0] : 0.0, warm1 ? wz[1] : 0.0}, y1[kSlots];
#pragma unroll
        for (int j = 0; j < kSlots; j++) y1[j] = 0.0;
        const bool c1 = quad_cone<1>(net, row, on1, st_gid, b, h, r1, z1, y1, warm1);
        const RowExact e1 = quad_exact_rows_worst(G, class_cap, net, q, m, row, st_gid, y1);
        const bool ok1 = c1 && e1.viol == 0u && e1.cap_viol == 0u;
        if (ok1) {
#pragma unroll
            for (int j = 0; j < kSlots; j++) y[j] = y1[j];
            keep = true;
            keep_row[0] = r1[0]; keep_row[1] = -1;
            keep_z[0] = z1[0]; keep_z[1] = z1[1];
        }
        settled = settled || ok1;
        go2 = c1 && !ok1 && e1.viol != 0u && e1.worst != r1[0] && e1.worst >= 0;
        if (go2) { pair[0] = r1[0]; pair[1] = e1.worst; zp[0] = z1[0]; zp[1] = z1[1]; }
    }
    // ONE two-row solve for every source of a pair (one inlined copy of quad_cone<2>: registers)
    const bool want2 = d2 || go2 || warm2;
    if (__ballot(want2) != 0ull) {
        const int r2[2] = {warm2 ? wrow[0] : pair[0], warm2 ? wrow[1] : pair[1]};
        double z2[4] = {warm2 ? wz[0] : zp[0], warm2 ? wz[1] : zp[1], warm2 ? wz[2] : 0.0, warm2 ? wz[3] : 0.0}, y2[kSlots];
#pragma unroll
        for (int j = 0; j < kSlots; j++) y2[j] = 0.0;
        const bool c2 = quad_cone<2>(net, row, want2, st_gid, b, h, r2, z2, y2, warm2);
        const RowExact e2 = quad_exact_rows_worst(G, class_cap, net, q, m, row, st_gid, y2);
        const bool ok2 = c2 && e2.viol == 0u && e2.cap_viol == 0u;
        if (ok2) {
#pragma unroll
            for (int j = 0; j < kSlots; j++) y[j] = y2[j];
            keep = true;
            keep_row[0] = r2[0]; keep_row[1] = r2[1];
#pragma unroll
            for (int a = 0; a < 4; a++) keep_z[a] = z2[a];
        }
        settled = settled || ok2;
    }
    // the multipliers the row settled with, for the next period
    const bool st = zw != nullptr && settled && keep;
    if (__ballot(st) != 0ull) {
        if (st && q < m) {
            double v0 = 0.0, v1 = 0.0;
            if ((int)q == keep_row[0]) { v0 = keep_z[0]; v1 = keep_z[1]; }
            if ((int)q == keep_row[1]) { v0 = keep_z[2]; v1 = keep_z[3]; }
            zw[q][0] = v0;
            zw[q][1] = v1;
        }
    }
    if (stored) *stored = st;
    // Tie snap: values the solver moved go to the 2^-16 A grid, exactly as solve_projection does it
    if (__ballot(settled) != 0ull) {
#pragma unroll
        for (int j = 0; j < kSlots; j++)
            if (settled && st_gid[j] >= 0 && y[j] != y0[j]) y[j] = tie_snap_counted(y[j], h[j], is_cc[j], tie_counters, tie_log2);
    }
    return settled;
}

}  // namespace evc
