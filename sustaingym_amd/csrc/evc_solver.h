// evc_solver.h — iterative action projection for the environments the main kernel queued.
//
// Problem (reference: env.py:178-221 + magnitude_constraint env.py:473-500, a cvxpy/MOSEK SOCP),
// in amps y = 32 x:
//     min 1/2 ||y - b||^2   s.t.  0 <= y <= h,   || M_c S(y) || <= r_c   (c < m)
// where S_g(y) = sum of y over station class g.  Conic dual with one 2-vector multiplier per row:
//     q(z) = min_{0<=y<=h} 1/2||y-b||^2 + sum_c ( z_c . M_c S(y) - r_c ||z_c|| )
// whose inner minimiser is the clip  y_i = clip(b_i - nu_g(i), 0, h_i),  nu = sum_c M_c^T z_c.
// q is concave with gradient  w_c - r_c z_c/||z_c||  (w_c = M_c S); z_c = 0 is optimal iff
// ||w_c|| <= r_c.  We ascend q with a Levenberg-Marquardt Newton direction on the active rows
// (Hessian  M_A diag(k_g) M_A^T + r_c/||z_c|| (I - z^ z^T), k_g = number of unclamped stations
// of class g) and a line search on the sign of the directional derivative (expand on flat
// pieces, halve on overshoot).  One "pass" = one clip + class sums; everything else is tiny
// dense algebra shared by the wave through LDS.
//
// One workgroup = one wavefront = one environment (lane i: station i; lane c: constraint row c;
// lane a: row a of the Newton system).
#pragma once

#include "evc_kernels.h"

namespace evc {

constexpr int kMaxActive = 8;               // rows simultaneously in the Newton system (extras wait; beyond that the
                                            // proximal-gradient safeguard finishes): keeps a workspace at 5 KB
constexpr int kMaxDim = 2 * kMaxActive;
constexpr int kSolverMaxIter = 60;

#ifdef EVC_SOLVER_STATS   // diagnostic builds only (tools/build_variant.sh): work statistics of the slow kernel
__device__ unsigned long long g_solver_stats[32];
#define SOLVER_CLK() clock64()
#define SOLVER_STAT(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_solver_stats[i], (unsigned long long)(v)); } while (0)
#else
#define SOLVER_STAT(i, v) do { } while (0)
#define SOLVER_CLK() 0ll
#endif

// One workgroup = one wavefront: LDS hand-offs between lanes need no s_barrier (the LDS pipeline
// executes a wave's ds instructions in issue order), only a compiler barrier.
#define SOLVER_SYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)

// Per-wavefront workspace of one solve; the network tables are shared by the workgroup's wavefronts.
struct SolverWs {
    double z[EVC_MAX_CONSTRAINTS][2];       // accepted multipliers
    double zt[EVC_MAX_CONSTRAINTS][2];      // trial multipliers
    double w[EVC_MAX_CONSTRAINTS][2];       // M_c S at the last pass
    double nu[EVC_MAX_GROUPS];
    double S[EVC_MAX_GROUPS];
    double kfree[EVC_MAX_GROUPS];
    double H[kMaxDim][kMaxDim + 1];
    double dir[kMaxDim];
    double zn[EVC_MAX_CONSTRAINTS];         // |z_c| and z_c / |z_c| of the active rows
    double zh[EVC_MAX_CONSTRAINTS][2];
    int act[kMaxActive];
};
// What a solve works on: the tables + its workspace (a view: references into LDS, resolved at compile time).
struct SolverLds {
    const LdsNet& net;
    double (&z)[EVC_MAX_CONSTRAINTS][2];
    double (&zt)[EVC_MAX_CONSTRAINTS][2];
    double (&w)[EVC_MAX_CONSTRAINTS][2];
    double (&nu)[EVC_MAX_GROUPS];
    double (&S)[EVC_MAX_GROUPS];
    double (&kfree)[EVC_MAX_GROUPS];
    double (&H)[kMaxDim][kMaxDim + 1];
    double (&dir)[kMaxDim];
    double (&zn)[EVC_MAX_CONSTRAINTS];
    double (&zh)[EVC_MAX_CONSTRAINTS][2];
    int (&act)[kMaxActive];
    __device__ __forceinline__ SolverLds(const LdsNet& n, SolverWs& ws)
        : net(n), z(ws.z), zt(ws.zt), w(ws.w), nu(ws.nu), S(ws.S), kfree(ws.kfree), H(ws.H), dir(ws.dir), zn(ws.zn), zh(ws.zh), act(ws.act) {}
};

struct SolverLane {
    double b, h;       // target / cap of this station (amps)
    int gid;           // station class of this lane (-1 outside the network)
    double y;          // current clip
};

// one station pass at multipliers zsrc: fills L.nu, L.S, L.kfree, L.w; lane.y
__device__ __forceinline__ void solver_pass(const Params& P, SolverLds& L, SolverLane& ln, int lane,
                                            double (*zsrc)[2]) {
    const int m = P.m, G = P.G;
    // rows that carry a multiplier at all (one or two, typically): nu sums over those only
    unsigned long long zrows = __ballot(lane < m && (zsrc[lane < m ? lane : 0][0] != 0.0 || zsrc[lane < m ? lane : 0][1] != 0.0));
    if (lane < G) {
        double nu = 0.0;
        while (zrows) {
            const int c = __builtin_ctzll(zrows);
            zrows &= zrows - 1ull;
            nu += L.net.Mre[lane][c] * zsrc[c][0] + L.net.Mim[lane][c] * zsrc[c][1];
        }
        L.nu[lane] = nu;
    }
    SOLVER_SYNC();
    double v = 0.0;
    bool is_free = false;
    ln.y = 0.0;
    if (ln.gid >= 0) {
        v = ln.b - L.nu[ln.gid];
        ln.y = fmin(fmax(v, 0.0), ln.h);
        is_free = (v > 0.0) && (v <= ln.h) && (ln.h > 0.0);
    }
    const unsigned long long free_mask = __ballot(is_free);
    for (int g0 = 0; g0 < G; g0 += 4) {            // four independent DPP ladders in flight (a lone one is all latency)
        double s[4];
#pragma unroll
        for (int u = 0; u < 4; u++) s[u] = wave_sum_f64(ln.gid == g0 + u ? ln.y : 0.0);
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (lane == 0 && g0 + u < G) {
                L.S[g0 + u] = s[u];
                L.kfree[g0 + u] = (double)__popcll(free_mask & P.group_mask[g0 + u]);
            }
    }
    SOLVER_SYNC();
    if (lane < m) {
        double re = 0.0, im = 0.0;
#pragma unroll 4
        for (int g = 0; g < G; g++) {             // unrolled: the LDS reads of a group are issued together
            re += L.net.Mre[g][lane] * L.S[g];
            im += L.net.Mim[g][lane] * L.S[g];
        }
        L.w[lane][0] = re;
        L.w[lane][1] = im;
    }
    SOLVER_SYNC();
}

// gradient of q for row c = lane at multipliers zsrc (0 for inactive rows)
__device__ __forceinline__ void row_gradient(const SolverLds& L, int m, int lane, double (*zsrc)[2],
                                             double& g0, double& g1, double& nz, double& nw) {
    g0 = g1 = nz = nw = 0.0;
    if (lane < m) {
        const double z0 = zsrc[lane][0], z1 = zsrc[lane][1];
        const double w0 = L.w[lane][0], w1 = L.w[lane][1];
        nz = sqrt(z0 * z0 + z1 * z1);
        nw = sqrt(w0 * w0 + w1 * w1);
        if (nz > 0.0) {
            g0 = w0 - L.net.mag[lane] * z0 / nz;
            g1 = w1 - L.net.mag[lane] * z1 / nz;
        }
    }
}

// trial point z + alpha * dir (rows whose multiplier would cross zero are deactivated); runs a
// pass there and returns the directional derivative per unit alpha.
__device__ __forceinline__ double solver_trial(const Params& P, SolverLds& L, SolverLane& ln, int lane,
                                               unsigned long long active, double alpha) {
    const int m = P.m;
    if (lane < m) {
        double t0 = L.z[lane][0], t1 = L.z[lane][1];
        if ((active >> lane) & 1ull) {
            const int j = __popcll(active & ((1ull << lane) - 1ull));
            const double z0 = t0, z1 = t1;
            t0 = z0 + alpha * L.dir[2 * j];
            t1 = z1 + alpha * L.dir[2 * j + 1];
            if (t0 * z0 + t1 * z1 <= 0.0) t0 = t1 = 0.0;
        }
        L.zt[lane][0] = t0;
        L.zt[lane][1] = t1;
    }
    SOLVER_SYNC();
    solver_pass(P, L, ln, lane, L.zt);
    double g0, g1, nz, nw;
    row_gradient(L, m, lane, L.zt, g0, g1, nz, nw);
    double dd = 0.0;
    if (lane < m && ((active >> lane) & 1ull))
        dd = g0 * (L.zt[lane][0] - L.z[lane][0]) + g1 * (L.zt[lane][1] - L.z[lane][1]);
    return wave_sum_f64(dd) / alpha;
}

__device__ __forceinline__ void accept_trial(SolverLds& L, int m, int lane) {
    if (lane < m) { L.z[lane][0] = L.zt[lane][0]; L.z[lane][1] = L.zt[lane][1]; }
    SOLVER_SYNC();
}

// In-LDS Cholesky solve of the d x d system (lane a owns row a and rhs a); result in L.dir.
__device__ __forceinline__ void solver_cholesky(SolverLds& L, int d, int lane, double rhs) {
    for (int j = 0; j < d; j++) {
        double t = 0.0;
        if (lane >= j && lane < d) {
            t = L.H[lane][j];
            for (int k = 0; k < j; k++) t -= L.H[lane][k] * L.H[j][k];
        }
        double piv = readlane_f64(t, j);
        piv = piv < 1e-300 ? 1e-300 : piv;
        const double ljj = sqrt(piv);
        SOLVER_SYNC();
        if (lane == j) L.H[j][j] = ljj;
        else if (lane > j && lane < d) L.H[lane][j] = t / ljj;
        SOLVER_SYNC();
    }
    // forward substitution L u = rhs
    for (int i = 0; i < d; i++) {
        double xi = readlane_f64(rhs, i) / L.H[i][i];
        if (lane == i) rhs = xi;
        else if (lane > i && lane < d) rhs -= L.H[lane][i] * xi;
    }
    // back substitution L^T x = u
    for (int i = d - 1; i >= 0; i--) {
        double xi = readlane_f64(rhs, i) / L.H[i][i];
        if (lane == i) rhs = xi;
        else if (lane < i) rhs -= L.H[i][lane] * xi;
    }
    if (lane < d) L.dir[lane] = rhs;
    SOLVER_SYNC();
}

// D x D system (D = 2, 4) solved redundantly by every lane in registers (Gauss elimination without
// pivoting: H is SPD plus the Levenberg-Marquardt shift): no LDS round trips, readlanes or barriers
// between the steps, which is what the in-LDS Cholesky spends its time on at these sizes.
template <int D>
__device__ __forceinline__ void solver_small(SolverLds& L, int lane, double rhs) {
    double A[D][D], x[D];
#pragma unroll
    for (int i = 0; i < D; i++) {
        x[i] = readlane_f64(rhs, i);
#pragma unroll
        for (int j = 0; j < D; j++) A[i][j] = L.H[i][j];
    }
    double inv[D];
#pragma unroll
    for (int k = 0; k < D; k++) {
        double piv = A[k][k];
        piv = piv < 1e-300 ? 1e-300 : piv;
        double rc0 = __builtin_amdgcn_rcp(piv);            // v_rcp_f64 + two Newton steps: the direction
        rc0 = rc0 * (2.0 - piv * rc0);                     // need not be correctly rounded
        inv[k] = rc0 * (2.0 - piv * rc0);
#pragma unroll
        for (int i = k + 1; i < D; i++) {
            const double f = A[i][k] * inv[k];
#pragma unroll
            for (int j = k + 1; j < D; j++) A[i][j] -= f * A[k][j];
            x[i] -= f * x[k];
        }
    }
#pragma unroll
    for (int i = D - 1; i >= 0; i--) {
        double acc = x[i];
#pragma unroll
        for (int j = i + 1; j < D; j++) acc -= A[i][j] * x[j];
        x[i] = acc * inv[i];
    }
    double mine = 0.0;
#pragma unroll
    for (int i = 0; i < D; i++) mine = lane == i ? x[i] : mine;
    if (lane < D) L.dir[lane] = mine;
    SOLVER_SYNC();
}

// The same for D = 6, 8 (three or four active rows: the usual case at Caltech's congested midday, where the in-LDS
// Cholesky took 39 k of a solve's 109 k cycles): the system is symmetric, so only the upper triangle is held
// (D (D + 1) / 2 doubles) — the multiplier of row i at elimination step k is U[k][i] / U[k][k].
template <int D>
__device__ __forceinline__ void solver_small_sym(SolverLds& L, int lane, double rhs) {
    double U[D][D], x[D], inv[D];
#pragma unroll
    for (int i = 0; i < D; i++) {
        x[i] = readlane_f64(rhs, i);
#pragma unroll
        for (int j = i; j < D; j++) U[i][j] = L.H[i][j];
    }
#pragma unroll
    for (int k = 0; k < D; k++) {
        double piv = U[k][k];
        piv = piv < 1e-300 ? 1e-300 : piv;
        double rc0 = __builtin_amdgcn_rcp(piv);
        rc0 = rc0 * (2.0 - piv * rc0);
        inv[k] = rc0 * (2.0 - piv * rc0);
#pragma unroll
        for (int i = k + 1; i < D; i++) {
            const double f = U[k][i] * inv[k];
#pragma unroll
            for (int j = i; j < D; j++) U[i][j] -= f * U[k][j];
            x[i] -= f * x[k];
        }
    }
#pragma unroll
    for (int i = D - 1; i >= 0; i--) {
        double acc = x[i];
#pragma unroll
        for (int j = i + 1; j < D; j++) acc -= U[i][j] * x[j];
        x[i] = acc * inv[i];
    }
    double mine = 0.0;
#pragma unroll
    for (int i = 0; i < D; i++) mine = lane == i ? x[i] : mine;
    if (lane < D) L.dir[lane] = mine;
    SOLVER_SYNC();
}

// Safeguard for the Newton ascent (it can stall where many overlapping rows make the multipliers non-unique:
// 2 of 30 random networks in tests/soak/network_fuzz.py): accelerated proximal gradient on the same dual.
// q(z) = f(z) - sum_c r_c ||z_c||, f(z) = min_{0<=y<=h} 1/2||y-b||^2 + z.By concave with gradient B y(z), Lipschitz
// with L <= P.prox_step^-1 (Gershgorin on B B', evc_engine.hip build_tables); the prox of the norm term is a block
// soft-threshold.  FISTA with gradient restart — globally convergent, 50-550 passes in practice
// (tools/proj_fallback_proto.py) — and only entered when the Newton did not deliver.  z = L.z, momentum point = L.zt.
// Leaves the pass state of the returned z; returns true if the Newton's KKT test is met.
__device__ __forceinline__ bool solver_proximal_gradient(const Params& P, SolverLds& L, SolverLane& ln, int lane) {
    const int m = P.m;
    const double t = P.prox_step;
    if (!(t > 0.0)) return false;
    if (lane < m) { L.z[lane][0] = 0.0; L.z[lane][1] = 0.0; L.zt[lane][0] = 0.0; L.zt[lane][1] = 0.0; }
    SOLVER_SYNC();
    double theta = 1.0;
    for (int k = 0; k < 50000; k++) {             // bounded: ~20 ms for one environment at worst
        solver_pass(P, L, ln, lane, L.zt);
        double zn0 = 0.0, zn1 = 0.0, z0 = 0.0, z1 = 0.0, rs = 0.0;
        if (lane < m) {
            const double v0 = L.zt[lane][0], v1 = L.zt[lane][1];
            const double u0 = v0 + t * L.w[lane][0], u1 = v1 + t * L.w[lane][1];
            const double nu = sqrt(u0 * u0 + u1 * u1);
            const double shrink = nu > 0.0 ? fmax(1.0 - t * L.net.mag[lane] / nu, 0.0) : 0.0;
            zn0 = u0 * shrink; zn1 = u1 * shrink;
            z0 = L.z[lane][0]; z1 = L.z[lane][1];
            rs = (zn0 - z0) * (v0 - zn0) + (zn1 - z1) * (v1 - zn1);
        }
        const double restart = wave_sum_f64(rs);
        double theta_n = 1.0, beta = 0.0;
        if (!(restart > 0.0)) {
            theta_n = 0.5 * (1.0 + sqrt(1.0 + 4.0 * theta * theta));
            beta = (theta - 1.0) / theta_n;
        }
        theta = theta_n;
        if (lane < m) {
            L.zt[lane][0] = zn0 + beta * (zn0 - z0); L.zt[lane][1] = zn1 + beta * (zn1 - z1);
            L.z[lane][0] = zn0; L.z[lane][1] = zn1;
        }
        SOLVER_SYNC();
        if ((k & 15) == 15) {
            solver_pass(P, L, ln, lane, L.z);
            double g0, g1, nz, nw;
            row_gradient(L, m, lane, L.z, g0, g1, nz, nw);
            bool bad = false;
            if (lane < m) {
                const double rc = L.net.mag[lane];
                bad = nz > 0.0 ? sqrt(g0 * g0 + g1 * g1) / rc > Consts::PROJ_TOL_KKT : nw / rc - 1.0 > Consts::PROJ_TOL;
            }
            if (__ballot(bad) == 0ull) return true;
        }
    }
    solver_pass(P, L, ln, lane, L.z);
    return false;
}

// ---- one or two cone rows decide: the Newton in registers ------------------------------------------------------------
// With the active rows known (R = 1: the most violated row; R = 2: it and the row violated at its optimum) the dual has
// D = 2R unknowns and every quantity of an iteration is a wave sum over the station lanes: w_a = sum c_a(i) y_i and
// K_ab = sum_{free i} c_a(i) c_b(i), with c(i) in R^D the rows' coefficients of station i's class.  No LDS exchange, no
// barriers, the D x D system solved redundantly by every lane — an iteration is a handful of interleaved DPP ladders
// instead of the general path's five LDS round trips (10 000 cycles per iteration there, measured).  Same start, same
// Levenberg-Marquardt shift and same KKT target as the general iteration, full steps only; anything irregular (a
// multiplier that would cross zero, the iteration budget) returns false and the general iteration below takes over
// from scratch.  On JPL's GMM middays 92 - 99 % of the queue is the R = 1 case, the rest almost all R = 2.
struct ExactRows { unsigned long long viol; unsigned cap_viol; int worst; };
// exact float64 rows of schedule y (four class ladders in flight) + the row furthest above its limit
__device__ __forceinline__ ExactRows exact_rows_worst(const Params& P, const LdsNet& net, const LaneNet& ln, int lane, double y) {
    double re = 0.0, im = 0.0;
    ExactRows out{0ull, 0u, -1};
    for (int g0 = 0; g0 < P.G; g0 += 4) {
        double S[4];
#pragma unroll
        for (int u = 0; u < 4; u++) S[u] = wave_sum_f64(ln.gid == g0 + u ? y : 0.0);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int g = g0 + u;
            if (g < P.G) {
                if (lane < P.m) { re += net.Mre[g][lane] * S[u]; im += net.Mim[g][lane] * S[u]; }
                if (S[u] > P.class_cap[g] * (1.0 + Consts::PROJ_TOL)) out.cap_viol |= 1u << g;
            }
        }
    }
    // squared ratios order the rows like the ratios do: no square root, one divide
    double ratio = 0.0;
    bool viol = false;
    if (lane < P.m) {
        const double mg = net.mag[lane], lim = mg * (1.0 + Consts::PROJ_TOL), m2 = re * re + im * im;
        viol = m2 > lim * lim;
        ratio = m2 / (mg * mg);
    }
    out.viol = __ballot(viol);
    if (out.viol != 0ull) {
        // rows live in lanes 0..31: maximum inside the two 16-lane DPP rows, then across them; the first lane that holds it
        double best = viol ? ratio : 0.0;
        best = fmax(best, dpp_f64<0x121, 0xf, false>(best));
        best = fmax(best, dpp_f64<0x122, 0xf, false>(best));
        best = fmax(best, dpp_f64<0x124, 0xf, false>(best));
        best = fmax(best, dpp_f64<0x128, 0xf, false>(best));
        best = fmax(readlane_f64(best, 0), readlane_f64(best, 16));
        out.worst = __builtin_ctzll(__ballot(viol && ratio == best) | (1ull << 63));
    }
    return out;
}

template <int R>
__device__ __forceinline__ bool wave_cone(const LdsNet& net, int gid, double b, double h, const int (&rows)[R],
                                          double (&z)[2 * R], double& yout, bool all_warm = false) {
    constexpr int D = 2 * R;
    double cf[D], rmag[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        cf[2 * r] = gid >= 0 ? net.Mre[gid][rows[r]] : 0.0;
        cf[2 * r + 1] = gid >= 0 ? net.Mim[gid][rows[r]] : 0.0;
        rmag[r] = net.mag[rows[r]];
    }
    double mu = 1e-3;
    // K_ab = sum over the FREE stations of c_a c_b depends on the multipliers only through which stations are free: it is
    // summed again only when that set changed since the last iteration (same values either way; a D = 6 iteration is 6 wave
    // sums instead of 27 once the clamp pattern has settled — a lone wavefront pays ~140 cycles per float64 wave sum,
    // tools/probes/prim_probe.hip)
    unsigned long long free_set = 0ull;
    bool have_K = false;
    double K[D][D];
    for (int it = 0; it < (R == 1 ? 14 : 24); it++) {
        double nu = 0.0;
#pragma unroll
        for (int a = 0; a < D; a++) nu += cf[a] * z[a];
        const double v = b - nu;
        const double y = gid >= 0 ? fmin(fmax(v, 0.0), h) : 0.0;
        const bool fr = gid >= 0 && (v > 0.0) && (v <= h) && (h > 0.0);
        double w[D];
#pragma unroll
        for (int a = 0; a < D; a++) w[a] = wave_sum_f64(cf[a] * y);
        const unsigned long long now_free = __ballot(fr);
        if (!have_K || now_free != free_set) {
#pragma unroll
            for (int a = 0; a < D; a++)
#pragma unroll
                for (int e = a; e < D; e++) K[a][e] = wave_sum_f64(fr ? cf[a] * cf[e] : 0.0);
            free_set = now_free;
            have_K = true;
        }
        if (it == 0 && !all_warm) {
            // the row without a multiplier: alone, the first-order size along its w; beside an active row, tiny
            const int a0 = D - 2;
            const double nw = sqrt(w[a0] * w[a0] + w[a0 + 1] * w[a0 + 1]);
            if (!(nw > rmag[R - 1])) return false;
            const double wh0 = w[a0] / nw, wh1 = w[a0 + 1] / nw;
            double lam = 1e-6;
            if (R == 1) {
                const double curv = wh0 * (K[0][0] * wh0 + K[0][1] * wh1) + wh1 * (K[0][1] * wh0 + K[1][1] * wh1);
                if (curv > 0.0) lam = fmax((nw - rmag[0]) / curv, 1e-6);
            }
            z[a0] = lam * wh0;
            z[a0 + 1] = lam * wh1;
            continue;
        }
        double zh[D], g[D], rn[R];
        bool conv = true;
#pragma unroll
        for (int r = 0; r < R; r++) {
            // one square root and one divide per row and iteration (the chain is latency: every IEEE divide is a dozen dependent instructions)
            const double inz = newton_rsqrt(z[2 * r] * z[2 * r] + z[2 * r + 1] * z[2 * r + 1]);
            zh[2 * r] = z[2 * r] * inz;
            zh[2 * r + 1] = z[2 * r + 1] * inz;
            g[2 * r] = w[2 * r] - rmag[r] * zh[2 * r];
            g[2 * r + 1] = w[2 * r + 1] - rmag[r] * zh[2 * r + 1];
            rn[r] = rmag[r] * inz;
            const double lim = Consts::PROJ_TOL_KKT * rmag[r];
            conv = conv && g[2 * r] * g[2 * r] + g[2 * r + 1] * g[2 * r + 1] <= lim * lim;
        }
        if (conv) { yout = y; return true; }
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
        // Gauss elimination without pivoting (SPD + shift)
        bool good = true;
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
        if (!good) return false;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const double t0 = z[2 * r] + d[2 * r], t1 = z[2 * r + 1] + d[2 * r + 1];
            if (!(t0 * z[2 * r] + t1 * z[2 * r + 1] > 0.0)) return false;     // the row would leave the active set
        }
#pragma unroll
        for (int a = 0; a < D; a++) z[a] += d[a];
        mu = fmax(mu * 0.25, 1e-12);
    }
    return false;
}

// Round 5 — ONE cone row beside class caps, with the caps ELIMINATED instead of carried as rows of their own (what (b2) below does
// with wave_cone<2|3>: a 4- or 6-dimensional Newton whose every iteration costs 6 float64 wave sums, 21 more whenever the clamp
// pattern moves, and a 6 x 6 elimination — 34 k cycles per Caltech midday residual, profiles/r5_solver_stats.txt, and the launch waits
// for it).  The relaxed problem  min |y - b|^2  s.t. box, every class cap, row c  has the KKT form
//     y_i = clip(b_i - M_g . z - nu_g, 0, h_i),   nu_g >= 0 only where class g sits AT its cap,   w = sum_g M_g S_g = limit z / |z|
// (M_g = the row's phasor coefficient of class g, z in R^2 the row's multiplier).  For a given z a capped class contributes exactly
// S_g = cap_g — its nu_g need not be known to evaluate w — and nothing to the Jacobian, so the Newton iteration runs on z alone
// (2 x 2), one wave sum per class of the row per iteration, the free counts from ballots; the caps' inner fillings are done once, at
// the end.  Same residual, shift and tolerance as wave_cone<1>; anything irregular returns false and the chain below takes over.
// yw = the schedule after the caps' filling (kept by classes the row does not load).  The caller verifies every row and cap.
__device__ __forceinline__ bool wave_cone_capped(const Params& P, const LdsNet& net, int gid, int c, double b, double h, double yw,
                                                 double& yout) {
    const int G = P.G;
    const double c0 = gid >= 0 ? net.Mre[gid][c] : 0.0, c1 = gid >= 0 ? net.Mim[gid][c] : 0.0;
    const bool in_row = c0 != 0.0 || c1 != 0.0;
    const double rmag = net.mag[c];
    // the classes the row loads, once: coefficient, cap, class id in registers (wave-uniform); a row over more than kRowClasses
    // classes is left to the chain below
    constexpr int kRowClasses = 6;
    double m0[kRowClasses], m1[kRowClasses], capv[kRowClasses];
    int cls[kRowClasses];
    int nc = 0;
#pragma unroll
    for (int j = 0; j < kRowClasses; j++) { m0[j] = 0.0; m1[j] = 0.0; capv[j] = HUGE_VAL; cls[j] = -1; }
    for (int g = 0; g < G; g++) {
        const double a0 = net.Mre[g][c], a1 = net.Mim[g][c];
        if (a0 == 0.0 && a1 == 0.0) continue;
        if (nc >= kRowClasses) return false;
#pragma unroll
        for (int j = 0; j < kRowClasses; j++)
            if (j == nc) { m0[j] = a0; m1[j] = a1; capv[j] = P.class_cap[g]; cls[j] = g; }
        nc++;
    }
    double z0 = 0.0, z1 = 0.0, mu = 1e-3;
    unsigned capped = 0u;
    for (int it = 0; it < 20; it++) {
        const double v = b - (c0 * z0 + c1 * z1);
        const double y = in_row ? fmin(fmax(v, 0.0), h) : 0.0;
        const bool fr = in_row && v > 0.0 && v <= h && h > 0.0;
        double w0 = 0.0, w1 = 0.0, K00 = 0.0, K01 = 0.0, K11 = 0.0;
        capped = 0u;
        double Wg[kRowClasses];
#pragma unroll
        for (int j = 0; j < kRowClasses; j++) Wg[j] = j < nc ? wave_sum_f64(gid == cls[j] ? y : 0.0) : 0.0;      // independent ladders
#pragma unroll
        for (int j = 0; j < kRowClasses; j++) {
            if (j >= nc) continue;
            double Sg = Wg[j];
            if (Wg[j] > capv[j]) { Sg = capv[j]; capped |= 1u << cls[j]; }
            else {
                const double kg = (double)__popcll(__ballot(fr && gid == cls[j]));
                K00 += kg * m0[j] * m0[j]; K01 += kg * m0[j] * m1[j]; K11 += kg * m1[j] * m1[j];
            }
            w0 += m0[j] * Sg; w1 += m1[j] * Sg;
        }
        if (it == 0) {                                   // first-order size along w (as wave_cone<1>)
            const double nw = sqrt(w0 * w0 + w1 * w1);
            if (!(nw > rmag)) return false;
            const double wh0 = w0 / nw, wh1 = w1 / nw;
            const double curv = wh0 * (K00 * wh0 + K01 * wh1) + wh1 * (K01 * wh0 + K11 * wh1);
            const double lam = curv > 0.0 ? fmax((nw - rmag) / curv, 1e-6) : 1e-6;
            z0 = lam * wh0; z1 = lam * wh1;
            continue;
        }
        const double inz = newton_rsqrt(z0 * z0 + z1 * z1);
        const double zh0 = z0 * inz, zh1 = z1 * inz;
        double g0 = w0 - rmag * zh0, g1 = w1 - rmag * zh1;
        const double lim = Consts::PROJ_TOL_KKT * rmag;
        if (g0 * g0 + g1 * g1 <= lim * lim) {
            double yy = in_row ? y : yw;
            [[maybe_unused]] const long long tf = SOLVER_CLK();
            for (int g = 0; g < G; g++)
                if ((capped >> g) & 1u) yy = waterfill_class(gid == g, v, h, P.class_cap[g], yy);
            yout = yy;
            SOLVER_STAT(24, SOLVER_CLK() - tf); SOLVER_STAT(25, __popc(capped));
            SOLVER_STAT(20, it);
            return true;
        }
        const double rn = rmag * inz;
        double B00 = K00 + rn * (1.0 - zh0 * zh0), B11 = K11 + rn * (1.0 - zh1 * zh1), B01 = K01 - rn * zh0 * zh1;
        double scale = 0.5 * (B00 + B11);
        scale = scale < 1e-12 ? 1e-12 : scale;
        B00 += mu * scale; B11 += mu * scale;
        const double det = B00 * B11 - B01 * B01;
        if (!(B00 > 0.0) || !(det > 0.0)) return false;
        const double idet = newton_rcp(det);
        const double d0 = (B11 * g0 - B01 * g1) * idet, d1 = (B00 * g1 - B01 * g0) * idet;
        const double t0 = z0 + d0, t1 = z1 + d1;
        if (!(t0 * z0 + t1 * z1 > 0.0)) return false;   // the row would leave the active set
        z0 = t0; z1 = t1;
        mu = fmax(mu * 0.25, 1e-12);
    }
    return false;
}

#ifndef EVC_SOLVE_ENV_INLINE
#define EVC_SOLVE_ENV_INLINE __forceinline__
#endif
// The projection of ONE environment by one wavefront: lane i holds station i's target ln.b = 32 a and cap ln.h (amps);
// returns the projected (and, where the solver moved it, tie-snapped) value of the lane's station.  `noconv` is set when
// neither the Newton iteration nor the proximal-gradient safeguard reached the tolerance (EVC_STATUS_PROJ_NOCONV).
// Shared by solve_env below (slow kernel, in-kernel drain) and by the fused rollout kernel (evc_rollout.h).
// Warm start (round 4, used by the fused rollout kernel): `warm` = the multipliers the general iteration ended with for THIS
// environment one period earlier ([m][2] in LDS), `warm->valid` says they are.  Under a policy whose actions hardly change from
// period to period (greedy) the dual optimum hardly moves either: the general iteration then starts beside it — the relaxation
// front (exact rows, filling, cone chain: ~50 000 cycles spent before the iteration even begins on a problem with many active
// rows) is skipped and two or three Newton iterations remain of seven.  The iteration activates and deactivates rows by itself
// (solver_trial), its convergence test covers every row, so a poor start costs time, never correctness.
struct SolveWarm { double (*z)[2]; bool valid; bool stored; };
__device__ __forceinline__ double solve_projection(const Params& P, SolverLds& L, const LaneNet& lnet, SolverLane& ln, int lane,
                                                   bool& noconv, SolveWarm* warm = nullptr) {
    const int m = P.m, G = P.G;
    noconv = false;
    [[maybe_unused]] long long c0 = SOLVER_CLK();
    const bool warm_start = warm != nullptr && warm->valid;
    if (lane < m) { L.z[lane][0] = warm_start ? warm->z[lane][0] : 0.0; L.z[lane][1] = warm_start ? warm->z[lane][1] : 0.0; }
    SOLVER_SYNC();
    if (warm) warm->stored = false;
    bool warm_settled = false;
#ifndef EVC_ABL_NO_WAVE_CONE
    if (warm_start) {
        // the rows that carried a multiplier one period ago, if there are at most three: the in-register Newton from those
        // multipliers (every row warm: no first-order start), accepted only if every row and cap holds at its optimum
        const unsigned long long act = __ballot(lane < m && (L.z[lane < m ? lane : 0][0] != 0.0 || L.z[lane < m ? lane : 0][1] != 0.0));
        const int na = __popcll(act);
        double yw = 0.0;
        bool got = false;
        if (na >= 1 && na <= 3) {
            int rr[3] = {0, 0, 0};
            double zz[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            unsigned long long a2 = act;
            for (int i = 0; i < na; i++) {
                rr[i] = __builtin_ctzll(a2);
                a2 &= a2 - 1ull;
                zz[2 * i] = L.z[rr[i]][0];
                zz[2 * i + 1] = L.z[rr[i]][1];
            }
            if (na == 1) { const int r1[1] = {rr[0]}; double z1[2] = {zz[0], zz[1]}; got = wave_cone<1>(L.net, lnet.gid, ln.b, ln.h, r1, z1, yw, true);
                           zz[0] = z1[0]; zz[1] = z1[1]; }
            else if (na == 2) { const int r2[2] = {rr[0], rr[1]}; double z2[4] = {zz[0], zz[1], zz[2], zz[3]}; got = wave_cone<2>(L.net, lnet.gid, ln.b, ln.h, r2, z2, yw, true);
                                for (int i = 0; i < 4; i++) zz[i] = z2[i]; }
            else { const int r3[3] = {rr[0], rr[1], rr[2]}; got = wave_cone<3>(L.net, lnet.gid, ln.b, ln.h, r3, zz, yw, true); }
            if (got) {
                const ExactRows ew = exact_rows_worst(P, L.net, lnet, lane, yw);
                if (ew.viol == 0ull && ew.cap_viol == 0u) {
                    ln.y = yw;
                    warm_settled = true;
                    if (lane == 0)
                        for (int i = 0; i < na; i++) { warm->z[rr[i]][0] = zz[2 * i]; warm->z[rr[i]][1] = zz[2 * i + 1]; }
                    warm->stored = true;
                    SOLVER_SYNC();
                }
            }
        }
    }
#endif

    // (a) the screen of the streaming kernel may have been merely inconclusive: exact test of
    //     the box clip; (b) class caps (pod breakers) violated: closed-form water-filling, exact if
    //     every row holds afterwards (relaxation argument)
    bool settled = warm_settled;
    if (!warm_start) {
        const double y0 = fmin(ln.b, ln.h);
        const ExactRows e0 = exact_rows_worst(P, L.net, lnet, lane, y0);
        [[maybe_unused]] const long long tA = SOLVER_CLK();
        SOLVER_STAT(15, tA - c0);
        ln.y = y0;
        if (e0.viol == 0ull) {
            settled = true;
        } else if (e0.cap_viol != 0u) {       // caps first, also beside violated multi-class rows
            double yw = y0;
            for (int g = 0; g < G; g++)
                if ((e0.cap_viol >> g) & 1u)
                    yw = waterfill_class(lnet.gid == g, ln.b, ln.h, P.class_cap[g], yw);
            const ExactRows ew = exact_rows_worst(P, L.net, lnet, lane, yw);
            [[maybe_unused]] const long long tB = SOLVER_CLK();
            SOLVER_STAT(16, tB - tA);
            if (ew.viol == 0ull) {
                ln.y = yw;
                settled = true;
            }
#ifndef EVC_ABL_NO_CONE_CAPPED
            else if (ew.worst >= 0) {
                // (b0) caps filled, a row still above its limit: that row with the caps eliminated (wave_cone_capped)
                double yc = 0.0;
                [[maybe_unused]] const long long tC = SOLVER_CLK();
                if (wave_cone_capped(P, L.net, lnet.gid, ew.worst, ln.b, ln.h, yw, yc)) {
                    SOLVER_STAT(26, SOLVER_CLK() - tC);
                    const ExactRows ec = exact_rows_worst(P, L.net, lnet, lane, yc);
                    if (ec.viol == 0ull && ec.cap_viol == 0u) {
                        ln.y = yc;
                        settled = true;
                    }
                }
                SOLVER_STAT(21, SOLVER_CLK() - tC); SOLVER_STAT(22, 1); SOLVER_STAT(23, settled ? 1 : 0);
            }
#endif
#ifndef EVC_ABL_NO_WAVE_CONE
            if (!settled && ew.viol != 0ull && __popc(e0.cap_viol) <= 2 && ew.worst >= 0) {
                // (b2) Caps filled, a row still violated (Caltech's congested middays: a feeder row beside the pod caps — one
                // such environment per step, and its solve is what the launch waits for).  The caps' own rows and the worst
                // remaining row, together and at once: the caps' multipliers are known from the filling (a class shifted by
                // nu_g has z_c = nu_g cf / |cf|^2 on its simple row c, cf = that row's coefficient of the class), the new row
                // starts as in wave_cone.  One in-register Newton on 2 - 3 rows instead of the chain 1 -> 2 -> 3 below.
                int rows_b[3] = {-1, -1, -1};
                double zb[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                int nc = 0;
                bool usable = true;
                for (int g = 0; g < G; g++) {
                    if (!((e0.cap_viol >> g) & 1u)) continue;
                    const unsigned long long rb = __ballot(lane < m && ((P.simple_rows >> lane) & 1u) &&
                                                           (L.net.Mre[g][lane] != 0.0 || L.net.Mim[g][lane] != 0.0));
                    const unsigned long long fb = __ballot(lnet.gid == g && yw > 0.0 && yw < ln.h && yw < ln.b);   // strictly inside: y = b - nu
                    if (rb == 0ull || fb == 0ull) { usable = false; break; }
                    const int c = __builtin_ctzll(rb);
                    const double nu = readlane_f64(ln.b - yw, __builtin_ctzll(fb));
                    const double cre = L.net.Mre[g][c], cim = L.net.Mim[g][c];
                    const double sc = nu / (cre * cre + cim * cim);
                    rows_b[nc] = c;
                    zb[2 * nc] = sc * cre;
                    zb[2 * nc + 1] = sc * cim;
                    nc++;
                }
                usable = usable && ew.worst != rows_b[0] && ew.worst != rows_b[1];
                double yb = 0.0;
                bool got = false;
                if (usable && nc == 1) {
                    const int r2[2] = {rows_b[0], ew.worst};
                    double z2[4] = {zb[0], zb[1], 0.0, 0.0};
                    got = wave_cone<2>(L.net, lnet.gid, ln.b, ln.h, r2, z2, yb);
                } else if (usable && nc == 2) {
                    const int r3[3] = {rows_b[0], rows_b[1], ew.worst};
                    double z3[6] = {zb[0], zb[1], zb[2], zb[3], 0.0, 0.0};
                    got = wave_cone<3>(L.net, lnet.gid, ln.b, ln.h, r3, z3, yb);
                }
                if (got) {
                    const ExactRows eb = exact_rows_worst(P, L.net, lnet, lane, yb);
                    if (eb.viol == 0ull && eb.cap_viol == 0u) {
                        ln.y = yb;
                        settled = true;
                    }
                }
                SOLVER_STAT(17, SOLVER_CLK() - tB);          // (b2): caps' rows + worst row at once
                SOLVER_STAT(18, 1); SOLVER_STAT(19, settled ? 1 : 0); SOLVER_STAT(20, nc);
            }
#endif
        }
#ifndef EVC_ABL_NO_WAVE_CONE
        if (!settled) {
            // (c) one to three rows decide: registers only, see wave_cone; exact if every row holds at the optimum.  A class
            // cap is a row like any other here (its own simple row of the matrix), so the chain also takes the environments
            // the water-filling left with a feeder row violated beside their pod caps (round 4: those went to the general
            // iteration below — ~84 000 cycles for 7 iterations on 2.6 rows, the tail of every congested Caltech step).
            const int r1[1] = {e0.worst};
            double z1[2] = {0.0, 0.0}, y1 = 0.0;
            if (wave_cone<1>(L.net, lnet.gid, ln.b, ln.h, r1, z1, y1)) {
                const ExactRows e1 = exact_rows_worst(P, L.net, lnet, lane, y1);
                if (e1.viol == 0ull && e1.cap_viol == 0u) {
                    ln.y = y1;
                    settled = true;
                } else if (e1.viol != 0ull && e1.worst != e0.worst) {
                    const int r2[2] = {e0.worst, e1.worst};
                    double z2[4] = {z1[0], z1[1], 0.0, 0.0}, y2 = 0.0;
                    if (wave_cone<2>(L.net, lnet.gid, ln.b, ln.h, r2, z2, y2)) {
                        const ExactRows e2 = exact_rows_worst(P, L.net, lnet, lane, y2);
                        if (e2.viol == 0ull && e2.cap_viol == 0u) {
                            ln.y = y2;
                            settled = true;
                        } else if (e2.viol != 0ull && e2.worst != r2[0] && e2.worst != r2[1]) {
                            // a third row: rare, but one such environment per step is what the whole launch waits for
                            const int r3[3] = {r2[0], r2[1], e2.worst};
                            double z3[6] = {z2[0], z2[1], z2[2], z2[3], 0.0, 0.0}, y3 = 0.0;
                            if (wave_cone<3>(L.net, lnet.gid, ln.b, ln.h, r3, z3, y3)) {
                                const ExactRows e3 = exact_rows_worst(P, L.net, lnet, lane, y3);
                                if (e3.viol == 0ull && e3.cap_viol == 0u) {
                                    ln.y = y3;
                                    settled = true;
                                }
                            }
                        }
                    }
                }
            }
        }
#endif
    }
    [[maybe_unused]] long long c1 = SOLVER_CLK();
    SOLVER_STAT(8, c1 - c0);
    [[maybe_unused]] long long t_build = 0, t_chol = 0, t_ls = 0, t_head = 0;
    SOLVER_STAT(0, 1);
    SOLVER_STAT(1, settled ? 1 : 0);
    if (!settled) solver_pass(P, L, ln, lane, L.z);
    [[maybe_unused]] int n_iter = 0, n_trial = 0, n_act = 0;

    double mu = 1e-3;
    bool converged = settled, last_ok = false;
    for (int it = 0; it < kSolverMaxIter && !settled; it++) {
        n_iter++;
        long long ca = SOLVER_CLK();
        double g0, g1, nz, nw;
        row_gradient(L, m, lane, L.z, g0, g1, nz, nw);
        const double rc = lane < m ? L.net.mag[lane] : 1.0;
        // activate violated rows that have no multiplier yet
        const bool newly = lane < m && nz == 0.0 && nw > rc * (1.0 + Consts::PROJ_TOL);
        const unsigned long long newly_rows = __ballot(newly);
        if (newly_rows != 0ull) {
            // First activation (no multiplier yet anywhere): start every violated row at the
            // first-order size of its multiplier along w — moving z_c by lam w^ lowers |w_c| by
            // about lam w^'(M_c diag(k) M_c')w^ — divided by the number of rows activated together,
            // whose corrections add up.  (A tiny start costs several expansion passes of the line
            // search per row: 12-15 passes per solve instead of 5; unscaled, a fully saturated
            // network overshoots into the flat region and cycles.)  Rows that become violated
            // later, beside active ones, start tiny and let the Newton system place them.
            const bool first = __ballot(lane < m && nz > 0.0) == 0ull;
            if (newly) {
                const double wh0 = L.w[lane][0] / nw, wh1 = L.w[lane][1] / nw;
                double lam = 1e-6;
                if (first) {
                    double curv = 0.0;
                    for (int g = 0; g < G; g++) {
                        const double pr = L.net.Mre[g][lane] * wh0 + L.net.Mim[g][lane] * wh1;
                        curv += L.kfree[g] * pr * pr;
                    }
                    if (curv > 0.0) lam = fmax((nw - rc) / (curv * (double)__popcll(newly_rows)), 1e-6);
                }
                L.z[lane][0] = lam * wh0;
                L.z[lane][1] = lam * wh1;
            }
            SOLVER_SYNC();
            solver_pass(P, L, ln, lane, L.z);
            row_gradient(L, m, lane, L.z, g0, g1, nz, nw);
        }
        unsigned long long active = __ballot(lane < m && nz > 0.0);
        // convergence test
        double res = 0.0;
        bool inact = true;
        if (lane < m) {
            inact = !(nz > 0.0);
            res = inact ? (nw / rc - 1.0) : sqrt(g0 * g0 + g1 * g1) / rc;
        }
        const unsigned long long bad_inact = __ballot(inact && res > Consts::PROJ_TOL);
        if (bad_inact == 0ull && __ballot(!inact && res > Consts::PROJ_TOL_KKT) == 0ull) {
            converged = true;
            break;
        }
        last_ok = bad_inact == 0ull && __ballot(!inact && res > Consts::PROJ_TOL_ACCEPT) == 0ull;
        // keep the Newton system within kMaxActive rows (drop the least violated extras)
        while (__popcll(active) > kMaxActive) {
            const int last = 63 - __clzll(active);
            active &= ~(1ull << last);
        }
        const int na = __popcll(active);
        n_act = na;
        long long cb = SOLVER_CLK(); t_head += cb - ca;
        const int d = 2 * na;
        const bool mine = lane < m && ((active >> lane) & 1ull);
        const int jrow = __popcll(active & ((1ull << lane) - 1ull));
        if (mine) L.act[jrow] = lane;
        SOLVER_SYNC();
        // Newton system, one ELEMENT per lane (d*d <= 1024 elements, row a = 2*j + p <-> active row
        // act[j], component p): H_ab = sum_g M_a[g] k_g M_b[g]  (+ the curvature of r_c ||z_c|| on the
        // 2x2 diagonal blocks); rhs = gradient of the active rows.
        if (mine) {
            const double z0 = L.z[lane][0], z1 = L.z[lane][1];
            const double nzc = sqrt(z0 * z0 + z1 * z1);
            L.zn[lane] = nzc; L.zh[lane][0] = z0 / nzc; L.zh[lane][1] = z1 / nzc;
        }
        SOLVER_SYNC();
        for (int e = lane; e < d * d; e += kWave) {
            const int a = e / d, bcol = e - a * d;
            const int ja = a >> 1, pa = a & 1, ca = L.act[ja];
            const int jb = bcol >> 1, pb = bcol & 1, cb = L.act[jb];
            double hsum = 0.0;
            for (int g = 0; g < G; g++) {
                const double ma = pa ? L.net.Mim[g][ca] : L.net.Mre[g][ca];
                const double mb = pb ? L.net.Mim[g][cb] : L.net.Mre[g][cb];
                hsum += ma * L.kfree[g] * mb;
            }
            if (jb == ja)
                hsum += (L.net.mag[ca] / L.zn[ca]) * ((pa == pb ? 1.0 : 0.0) - L.zh[ca][pa] * L.zh[ca][pb]);
            L.H[a][bcol] = hsum;
        }
        double rhs = 0.0;
        if (lane < d) {
            const int ca = L.act[lane >> 1], pa = lane & 1;
            rhs = L.w[ca][pa] - L.net.mag[ca] * L.zh[ca][pa];
        }
        SOLVER_SYNC();
        double tr = wave_sum_f64(lane < d ? L.H[lane][lane] : 0.0);
        double scale = tr / (double)d;
        scale = scale < 1e-12 ? 1e-12 : scale;
        if (lane < d) L.H[lane][lane] += mu * scale;
        SOLVER_SYNC();
        const double grad_a = rhs;
        long long cc = SOLVER_CLK(); t_build += cc - cb;
        if (d == 2) solver_small<2>(L, lane, rhs);          // one active row: the usual case
        else if (d == 4) solver_small<4>(L, lane, rhs);
        else if (d == 6) solver_small_sym<6>(L, lane, rhs);
        else if (d == 8) solver_small_sym<8>(L, lane, rhs);
        else solver_cholesky(L, d, lane, rhs);
        const double dd0 = wave_sum_f64(lane < d ? grad_a * L.dir[lane] : 0.0);

        long long cd = SOLVER_CLK(); t_chol += cd - cc;
        // line search on the sign of the directional derivative
        double alpha = 1.0;
        double dd = solver_trial(P, L, ln, lane, active, alpha);
        n_trial++;
        bool state_current = true;            // L.w / L.S / L.kfree / ln.y belong to the accepted point
        if (dd > 0.25 * dd0) {
            // undershoot (flat piece): expand while the derivative stays positive
            accept_trial(L, m, lane);         // alpha = 1 is an ascent point
            double best_alpha = 1.0;
            // z was overwritten: trials are taken relative to the ORIGINAL point, so keep
            // the displacement bookkeeping simple by expanding from the accepted point.
            while (dd > 0.25 * dd0 && best_alpha < 1e6) {
                const double dd2 = solver_trial(P, L, ln, lane, active, 3.0 * best_alpha);
                n_trial++;
                if (dd2 < -0.5 * dd0) { state_current = false; break; }     // rejected trial
                accept_trial(L, m, lane);
                best_alpha *= 4.0;
                dd = dd2;
            }
            mu = fmax(mu * 0.1, 1e-12);
        } else {
            int nback = 0;
            while (dd < -0.5 * dd0 && alpha > 1e-8) {
                alpha *= 0.5;
                nback++;
                dd = solver_trial(P, L, ln, lane, active, alpha);
                n_trial++;
            }
            accept_trial(L, m, lane);
            mu = (nback > 1) ? mu * 4.0 : fmax(mu * 0.25, 1e-12);
        }
        // the pass of the last trial already left the state of the accepted point, unless that
        // trial was rejected
        if (!state_current) solver_pass(P, L, ln, lane, L.z);
        t_ls += SOLVER_CLK() - cd;
    }
    [[maybe_unused]] long long c2 = SOLVER_CLK();
    SOLVER_STAT(9, c2 - c1); SOLVER_STAT(10, t_head); SOLVER_STAT(11, t_build); SOLVER_STAT(12, t_chol); SOLVER_STAT(13, t_ls);
    SOLVER_STAT(2, n_iter); SOLVER_STAT(3, n_trial); SOLVER_STAT(4, n_act); SOLVER_STAT(5, n_iter >= 20 ? 1 : 0);
    SOLVER_STAT(6, n_act == 1 ? 1 : 0); SOLVER_STAT(7, (!converged && !last_ok) ? 1 : 0);
    if (!converged && !last_ok) {                 // rare: the Newton stalled — globally convergent fallback
        converged = solver_proximal_gradient(P, L, ln, lane);
        if (!converged) noconv = true;
    }
    if (warm && !settled && !noconv) {            // the general iteration ran: its multipliers are next period's start
        // (a start settled by the in-register Newton above has stored its own)
        if (lane < m) { warm->z[lane][0] = L.z[lane][0]; warm->z[lane][1] = L.z[lane][1]; }
        warm->stored = true;
        SOLVER_SYNC();
    }
    // Tie snap: values the solver moved are snapped to a 2^-16 A grid so that
    // optima sitting exactly on a rounding boundary of env.py:373-378 round deterministically.
    double y = ln.y;
    if (y != fmin(ln.b, ln.h)) y = tie_snap_counted(y, ln.h, lnet.is_cc, P.tie_counters, P.tie_log2);
    return y;
}

// The projection + the rest of the step for ONE queued environment, by one wavefront (lane i: station i;
// lane c: constraint row c; lane a: row a of the Newton system).  Shared by the slow kernel and by the
// streaming kernel's in-kernel queue drain (evc_cquad.h).
template <int WORDS>
__device__ EVC_SOLVE_ENV_INLINE void solve_env(const Params& P, const StepIO& io, SolverLds& L, int lane, int env, int wave_in_block = 0) {
    const LaneNet lnet = lane_net(P, lane);
    SolverLane ln;
    ln.gid = lnet.gid;
    const EnvLoads cur = issue_loads(P, io, env, lane, wave_in_block);
    EnvRegs r;
    unpack_env(cur, r);
    bool clamped;
    const double a = unpack_action(io, cur, clamped);
    ln.b = a * Consts::ACTION_SCALE_FACTOR;
    ln.h = demand_cap_amps(r);
    bool noconv;
    const double y = solve_projection(P, L, lnet, ln, lane, noconv);
    if (noconv) r.status |= EVC_STATUS_PROJ_NOCONV;
    [[maybe_unused]] long long c2 = SOLVER_CLK();
    finish_step<WORDS>(P, io, L.net, lnet, env, lane, y, clamped, cur.acc, false, r);
    SOLVER_STAT(14, SOLVER_CLK() - c2);
    SOLVER_SYNC();
}

// Four wavefronts per workgroup, one queued environment each at a time, sharing the network tables in LDS
// (13 KB + 4 x 5 KB): EVC_SOLVER_WAVES workgroups per CU are resident, i.e. that many solves per SIMD in flight.  A solve
// is a chain of dependent float64 ladders, LDS round trips, square roots and divides on one wavefront — all latency —
// so the slow path's throughput is the number of wavefronts in flight (round 1/2: one 64-thread workgroup per solve,
// 25 KB of LDS and 250 VGPRs each, 6 per CU).  Round 3, with each half's slow launch running under the other half's
// streaming launch (pipelined halves): 2 per SIMD (256 VGPRs, 4 spilled instead of 83) beats 3 on JPL's GMM days — 47.8 - 48.4
// against 48.8 - 49.6 us per step pipelined, 52.3 - 53.2 against 55.1 - 56.2 as one launch; 4 (128 VGPRs, 173 spilled): 54.0.
#ifndef EVC_SOLVER_WAVES
#define EVC_SOLVER_WAVES 2
#endif
template <int WORDS>
__global__ __launch_bounds__(256, EVC_SOLVER_WAVES) void solver_step_kernel(Params P, StepIO io) {
    __shared__ LdsNet net;
    __shared__ SolverWs ws[4];
    const int count = rfl(*P.slow_count);
    if (blockIdx.x == 0 && threadIdx.x == 0) queue_begin_drain(P, count);
    if ((int)blockIdx.x * 4 >= count) return;       // nothing queued for this workgroup
    stage_net(net, P);
    const int lane = threadIdx.x & 63, wave = rfl((int)(threadIdx.x >> 6));
    SolverLds L(net, ws[wave]);
    for (int q = blockIdx.x * 4 + wave; q < count; q += gridDim.x * 4) solve_env<WORDS>(P, io, L, lane, rfl(P.slow_list[q]), wave);
}

}  // namespace evc
