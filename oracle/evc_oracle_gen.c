/*
 * evc_oracle_gen.c — CPU ORACLE (test infrastructure, NOT product code): episode generation.
 *
 * Scalar restatement of GMMsTraceGenerator._create_events / _sample
 * (sustaingym/envs/evcharging/event_generation.py:416-515) on a COUNTER-BASED random stream,
 * the specification the HIP kernel (sustaingym_amd/csrc/evc_gen.h) is checked against bit for
 * bit.  The reference draws from numpy Generator / sklearn RandomState streams, which cannot
 * be reproduced by thousands of independent GPU wavefronts; this generator keeps the
 * reference's MODEL (daily session count ~ empirical counts :479, iid GMM draws with the
 * rejection rules :441-455, arrival sort :490, availability-weighted EVSE choice :499-511,
 * dropped when no EVSE is free :514, random day :117-119) and replaces only the stream.
 * PARITY STATUS: distribution pinned against the golden-pinned numpy restatement of the
 * reference generator (tests/test_generator_oracle.py); seed-for-seed parity with the reference
 * is impossible by construction and not claimed.
 *
 * Random stream: Philox4x32-10 (Salmon et al., SC'11), key = 64-bit seed,
 * counter = (index, stream, episode_lo, episode_hi):
 *   stream 0, index 0      : word0 -> daily-count index, word1 -> day
 *   stream 1, index q      : candidate q: word0 -> mixture component, words1..3 -> z0..z2
 *   stream 2, index q      : candidate q: word0 -> z3     (q = round * m + i, m = int(1.2 n))
 *   stream 3, index j      : j-th session (arrival order): word0 -> EVSE choice
 * uniform(w) = (w + 0.5) * 2^-32; integers in [0,R) as (w * R) >> 32; standard normals by
 * Acklam's rational inverse CDF with a polynomial log (only + - * / sqrt, no libm, so that
 * gcc and the GPU produce identical doubles under -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "evc_oracle.h"

void orc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                    uint32_t out[4]) {
    for (int round = 0; round < 10; round++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double uniform32(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }

/* natural log for x in (0, 1]: x = m 2^e, m in [sqrt(1/2), sqrt 2); log m = 2 atanh(s),
 * s = (m-1)/(m+1), odd series to s^19 */
double orc_gen_log(double x) {
    uint64_t bits; memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m; memcpy(&m, &bits, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    const double f = m - 1.0, s = f / (2.0 + f), z = s * s;
    double p = 1.0 / 19.0;
    p = p * z + 1.0 / 17.0; p = p * z + 1.0 / 15.0; p = p * z + 1.0 / 13.0; p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;  p = p * z + 1.0 / 7.0;  p = p * z + 1.0 / 5.0;  p = p * z + 1.0 / 3.0;
    p = p * z;
    return (double)e * 0.6931471805599453 + (2.0 * s + 2.0 * s * p);
}

/* Acklam's inverse normal CDF (relative error 1.15e-9) */
double orc_gen_normal(double u) {
    static const double a[6] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                                1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[5] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                                6.680131188771972e+01, -1.328068155288572e+01};
    static const double c[6] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                                -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    static const double d[4] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00,
                                3.754408661907416e+00};
    const double plow = 0.02425;
    const double t = u < 1.0 - u ? u : 1.0 - u;          /* tail probability */
    if (t < plow) {
        const double q = sqrt(-2.0 * orc_gen_log(t));
        const double num = ((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5];
        const double den = (((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.0;
        const double x = num / den;                        /* negative */
        return u < 1.0 - u ? x : -x;
    }
    const double q = u - 0.5, r = q * q;
    const double num = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q;
    const double den = ((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0;
    return num / den;
}

#define GEN_MAX_BATCHES 8
#define GEN_MAX_BATCH 192

int orc_generate_episode(const orc_gmm* g, int n_stations, uint64_t seed, uint64_t episode,
                         int max_sessions, orc_session* sessions, double* requested,
                         int32_t* moer_day, double* max_profit) {
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint32_t e0 = (uint32_t)episode, e1 = (uint32_t)(episode >> 32);
    uint32_t w[4];
    orc_philox4x32(0, 0, e0, e1, k0, k1, w);
    int want = g->daily_counts[((uint64_t)w[0] * (uint64_t)g->n_counts) >> 32];      /* :479 */
    *moer_day = (int32_t)(((uint64_t)w[1] * (uint64_t)g->num_days) >> 32);          /* :117-119 */
    if (want < 0) want = 0;
    if (want > max_sessions) want = max_sessions;
    if (want > 128) want = 128;

    int arr[128], dep[128], est[128];
    double req[128];
    int have = 0;
    /* :416-463.  The reference draws int(n * 1.2) samples per round from sklearn's
     * GaussianMixture.sample, which returns them STACKED IN COMPONENT ORDER, filters them and
     * keeps the first n — so the surplus is cut from the highest-numbered components.  Kept:
     * each round draws m candidates, visits them in (component, index) order and accepts until
     * n sessions exist. */
    const int m = (int)((double)want * (1.0 + 0.2));
    for (int batch = 0; batch < GEN_MAX_BATCHES && have < want; batch++) {
        int c_comp[GEN_MAX_BATCH], c_t0[GEN_MAX_BATCH], c_t1[GEN_MAX_BATCH], c_t2[GEN_MAX_BATCH];
        double c_req[GEN_MAX_BATCH];
        unsigned char c_ok[GEN_MAX_BATCH];
        for (int i = 0; i < m; i++) {
            const uint32_t q = (uint32_t)(batch * m + i);
            uint32_t wa[4], wb[4];
            orc_philox4x32(q, 1, e0, e1, k0, k1, wa);
            orc_philox4x32(q, 2, e0, e1, k0, k1, wb);
            const double uc = uniform32(wa[0]);
            int comp = 0;
            for (int j = 0; j < g->n_components - 1; j++) comp += uc >= g->cum_weights[j];
            const double z[4] = {orc_gen_normal(uniform32(wa[1])), orc_gen_normal(uniform32(wa[2])),
                                 orc_gen_normal(uniform32(wa[3])), orc_gen_normal(uniform32(wb[0]))};
            double x[4];
            for (int r = 0; r < 4; r++) {
                double acc = g->means[comp * 4 + r];
                for (int col = 0; col <= r; col++) acc = acc + g->chol[comp * 16 + r * 4 + col] * z[col];
                x[r] = acc;
            }
            c_comp[i] = comp;
            c_ok[i] = 0;
            if (!(0.0 <= x[0] && x[1] < 1.0 && x[2] < 1.0 && x[3] >= 0.0)) continue;   /* :441-444 */
            const double t0 = floor(1440.0 * x[0] / 5.0), t1 = floor(1440.0 * x[1] / 5.0),
                         t2 = floor(1440.0 * x[2] / 5.0);                                /* :447-449 */
            if (!(t0 < t1 && t0 < t2)) continue;                                         /* :452-455 */
            double r = x[3] * 100.0;                                                     /* :458 */
            if (r < 0.0) r = 0.0;
            if (r > g->requested_energy_cap) r = g->requested_energy_cap;                /* :486 */
            c_ok[i] = 1; c_t0[i] = (int)t0; c_t1[i] = (int)t1; c_t2[i] = (int)t2; c_req[i] = r;
        }
        for (int comp = 0; comp < g->n_components && have < want; comp++)
            for (int i = 0; i < m && have < want; i++)
                if (c_ok[i] && c_comp[i] == comp) {
                    arr[have] = c_t0[i]; dep[have] = c_t1[i]; est[have] = c_t2[i]; req[have] = c_req[i];
                    have++;
                }
    }
    /* :490 sort by arrival (ties in draw order) */
    int order[128];
    for (int i = 0; i < have; i++) {
        int rank = 0;
        for (int j = 0; j < have; j++) rank += (arr[j] * 256 + j) < (arr[i] * 256 + i);
        order[rank] = i;
    }
    /* :499-511 EVSE choice among the free ones, weighted by historical usage */
    int station_dep[ORC_MAX_STATIONS];
    for (int s = 0; s < n_stations; s++) station_dep[s] = -1;
    int n_out = 0;
    double profit = 0.0;
    memset(sessions, 0, sizeof(orc_session) * (size_t)max_sessions);
    memset(requested, 0, sizeof(double) * (size_t)max_sessions);
    for (int j = 0; j < have; j++) {
        const int i = order[j];
        uint64_t total = 0;
        int n_avail = 0;
        for (int s = 0; s < n_stations; s++)
            if (station_dep[s] < arr[i]) { total += g->station_usage[s]; n_avail++; }
        if (n_avail == 0) continue;                                                  /* :502-503,514 */
        const int uniform_pick = total == 0;                                         /* :505-506 */
        if (uniform_pick) total = (uint64_t)n_avail;
        orc_philox4x32((uint32_t)j, 3, e0, e1, k0, k1, w);
        const uint64_t target = ((uint64_t)w[0] * total) >> 32;
        uint64_t run = 0;
        int pick = -1;
        for (int s = 0; s < n_stations && pick < 0; s++)
            if (station_dep[s] < arr[i]) {
                run += uniform_pick ? 1u : g->station_usage[s];
                if (run > target) pick = s;
            }
        if (dep[i] > station_dep[pick]) station_dep[pick] = dep[i];                  /* :510 */
        sessions[n_out].arrival = (int16_t)arr[i];
        sessions[n_out].departure = (int16_t)dep[i];
        sessions[n_out].est_departure = (int16_t)est[i];
        sessions[n_out].station = (int16_t)pick;
        requested[n_out] = req[i];
        const double cap_kwh = (double)(dep[i] - arr[i]) * 32.0 * ((1.0 / 60.0) * (208.0 / 1000.0) * 5.0);
        profit += (req[i] < cap_kwh ? req[i] : cap_kwh) * (0.15 * 0.20);              /* env.py:422-429 */
        n_out++;
    }
    *max_profit = profit;
    return n_out;
}

/* Device-resident RandomAlgorithm policy (reference: algorithms/evcharging/baselines.py:38-51 draws
 * rng.random(n) / rng.choice(5, n) from np.random.default_rng(); the engine replaces the sequential
 * stream by a counter-based one, include/evcharge.h:evc_set_policy_seed).  Action of every station of
 * global environment `env` in period t of its episode number `episode`:
 *   Philox4x32-10, key = seed, counter = (t | block << 16, episode, env, 0x504f4c43), block = s / 4,
 *   word s % 4;  continuous: (w >> 8) * 2^-24;  discrete: level = (w * bins) >> 32, a = level/(bins-1)
 *   in float32 (wrappers.py:43-45). */
void orc_random_action(uint64_t seed, uint32_t env, uint32_t episode, uint32_t t, int n, int bins,
                       float* out) {
    for (int s = 0; s < n; s++) {
        uint32_t w[4];
        orc_philox4x32(t | ((uint32_t)(s / 4) << 16), episode, env, 0x504f4c43u, (uint32_t)seed,
                       (uint32_t)(seed >> 32), w);
        const uint32_t x = w[s % 4];
        if (bins >= 2)
            out[s] = (float)(uint32_t)(((uint64_t)x * (uint32_t)bins) >> 32) / (float)(bins - 1);
        else
            out[s] = (float)(x >> 8) * (1.0f / 16777216.0f);
    }
}
