// evc_gen.h — on-device episode generation: the reset-time work of
// GMMsTraceGenerator._create_events (sustaingym/envs/evcharging/event_generation.py:416-515)
// as one kernel, so that refilling the episode bank at an autoreset boundary never touches the
// host.  One wavefront builds one episode:
//   header      daily session count n ~ empirical counts (:479), MOER day ~ U{0..days-1} (:117-119)
//   candidates  rounds of m = int(1.2 n) GMM draws, one per lane (Philox4x32-10 counter stream,
//               Acklam inverse normal CDF, x = mean_c + L_c z), rejection rules :441-455
//   accept      in (component, index) order — sklearn's GaussianMixture.sample stacks its draws by
//               component and the reference keeps the first n, so the surplus is cut from the
//               last components; rank by counting keys in LDS
//   sort        by arrival (:490), ties in accept order — rank by counting
//   EVSEs       lanes = stations: per session a DPP prefix sum of the usage counts of the free
//               EVSEs and one ballot pick the station (:499-511); no free EVSE drops it (:514)
//   store       compacted sessions / requested kWh / n_sessions / MOER day / max_profit to the bank
// The random stream and every arithmetic step are specified in DESIGN.md §9; an independent scalar
// C restatement of that specification reproduces the bank bit for bit (tests/test_gpu_generator.py).
#pragma once

#include "evc_device.h"

namespace evc {

constexpr int kGenMaxBatch = 192;      // >= int(1.2 * 128)
constexpr int kGenMaxBatches = 8;
constexpr int kGenMaxSessions = 128;

struct GenTables {
    int K, n_counts, num_days, pad;
    double cap;
    double cum[EVC_MAX_GMM_COMPONENTS];
    double means[EVC_MAX_GMM_COMPONENTS * 4];
    double chol[EVC_MAX_GMM_COMPONENTS * 16];
    int counts[EVC_MAX_DAILY_COUNTS];
    unsigned usage[EVC_MAX_STATIONS];
};

struct GenLds {
    double c_req[kGenMaxBatch];
    double a_req[kGenMaxSessions], s_req[kGenMaxSessions];
    unsigned rnd[kGenMaxSessions];
    unsigned short c_key[kGenMaxBatch];
    short c_t0[kGenMaxBatch], c_t1[kGenMaxBatch], c_t2[kGenMaxBatch];
    short a_arr[kGenMaxSessions], a_dep[kGenMaxSessions], a_est[kGenMaxSessions];
    short s_arr[kGenMaxSessions], s_dep[kGenMaxSessions], s_est[kGenMaxSessions], s_station[kGenMaxSessions];
};

__device__ __forceinline__ double gen_uniform(unsigned w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }

// log(x), x in (0,1]: x = m 2^e, m in [sqrt(1/2), sqrt 2), log m = 2 atanh((m-1)/(m+1)) to s^19.
// Only + - * / : the same doubles as any IEEE host compiler under -ffp-contract=off.
__device__ __forceinline__ double gen_log(double x) {
    unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((long long)((bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    const double f = m - 1.0, s = f / (2.0 + f), z = s * s;
    double p = 1.0 / 19.0;
    p = p * z + 1.0 / 17.0; p = p * z + 1.0 / 15.0; p = p * z + 1.0 / 13.0; p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;  p = p * z + 1.0 / 7.0;  p = p * z + 1.0 / 5.0;  p = p * z + 1.0 / 3.0;
    p = p * z;
    return (double)e * 0.6931471805599453 + (2.0 * s + 2.0 * s * p);
}

// Acklam's rational approximation of the inverse normal CDF.
__device__ __forceinline__ double gen_normal(double u) {
    const double lo = u < 1.0 - u ? u : 1.0 - u;
    if (lo < 0.02425) {
        const double q = __builtin_sqrt(-2.0 * gen_log(lo));
        const double num = ((((-7.784894002430293e-03 * q + -3.223964580411365e-01) * q + -2.400758277161838e+00) * q +
                             -2.549732539343734e+00) * q + 4.374664141464968e+00) * q + 2.938163982698783e+00;
        const double den = (((7.784695709041462e-03 * q + 3.224671290700398e-01) * q + 2.445134137142996e+00) * q +
                            3.754408661907416e+00) * q + 1.0;
        const double x = num / den;
        return u < 1.0 - u ? x : -x;
    }
    const double q = u - 0.5, r = q * q;
    const double num = (((((-3.969683028665376e+01 * r + 2.209460984245205e+02) * r + -2.759285104469687e+02) * r +
                          1.383577518672690e+02) * r + -3.066479806614716e+01) * r + 2.506628277459239e+00) * q;
    const double den = ((((-5.447609879822406e+01 * r + 1.615858368580409e+02) * r + -1.556989798598866e+02) * r +
                         6.680131188771972e+01) * r + -1.328068155288572e+01) * r + 1.0;
    return num / den;
}

// inclusive prefix sum over the wave (same DPP ladder as wave_scan_f64)
__device__ __forceinline__ unsigned wave_scan_u32(unsigned v) {
    v += dpp_u32<0x111, 0xf, true>(v);
    v += dpp_u32<0x112, 0xf, true>(v);
    v += dpp_u32<0x114, 0xf, true>(v);
    v += dpp_u32<0x118, 0xf, true>(v);
    v += dpp_u32<0x142, 0xa, false>(v);
    v += dpp_u32<0x143, 0xc, false>(v);
    return v;
}

__global__ __launch_bounds__(64) void generate_kernel(const GenTables* __restrict__ T, int n_stations, int stride,
                                                      evc_session* __restrict__ sessions,
                                                      double* __restrict__ requested, int* __restrict__ n_sessions,
                                                      int* __restrict__ slot_moer, double* __restrict__ max_profit,
                                                      int first_slot, int count, unsigned long long seed,
                                                      unsigned long long first_episode) {
    __shared__ GenLds L;
    const int lane = threadIdx.x;
    const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    const unsigned usage = lane < n_stations ? T->usage[lane] : 0u;
    const int K = T->K;
    const double cap = T->cap;

    for (int ep = blockIdx.x; ep < count; ep += gridDim.x) {
        const unsigned long long episode = first_episode + (unsigned long long)ep;
        const unsigned e0 = (unsigned)episode, e1 = (unsigned)(episode >> 32);
        const size_t slot = (size_t)(first_slot + ep);

        const Philox hdr(0, 0, e0, e1, k0, k1);
        int want = T->counts[(unsigned)(((unsigned long long)hdr.w[0] * (unsigned long long)T->n_counts) >> 32)];
        const int day = (int)(((unsigned long long)hdr.w[1] * (unsigned long long)T->num_days) >> 32);
        want = max(0, min(want, min(stride, kGenMaxSessions)));
        const int m = (int)((double)want * (1.0 + 0.2));

        int have = 0;
        for (int batch = 0; batch < kGenMaxBatches && have < want; batch++) {
            for (int i = lane; i < m; i += 64) {
                const unsigned q = (unsigned)(batch * m + i);
                const Philox wa(q, 1, e0, e1, k0, k1), wb(q, 2, e0, e1, k0, k1);
                const double uc = gen_uniform(wa.w[0]);
                int comp = 0;
                for (int j = 0; j < K - 1; j++) comp += uc >= T->cum[j];
                const double z0 = gen_normal(gen_uniform(wa.w[1])), z1 = gen_normal(gen_uniform(wa.w[2])),
                             z2 = gen_normal(gen_uniform(wa.w[3])), z3 = gen_normal(gen_uniform(wb.w[0]));
                const double* mu = T->means + comp * 4;
                const double* c = T->chol + comp * 16;
                const double x0 = mu[0] + c[0] * z0;
                const double x1 = (mu[1] + c[4] * z0) + c[5] * z1;
                const double x2 = ((mu[2] + c[8] * z0) + c[9] * z1) + c[10] * z2;
                const double x3 = (((mu[3] + c[12] * z0) + c[13] * z1) + c[14] * z2) + c[15] * z3;
                bool ok = 0.0 <= x0 && x1 < 1.0 && x2 < 1.0 && x3 >= 0.0;
                const double t0 = floor(1440.0 * x0 / 5.0), t1 = floor(1440.0 * x1 / 5.0),
                             t2 = floor(1440.0 * x2 / 5.0);
                ok = ok && t0 < t1 && t0 < t2;
                L.c_key[i] = ok ? (unsigned short)(comp * 256 + i) : (unsigned short)0xffff;
                if (ok) {
                    L.c_t0[i] = (short)(int)t0; L.c_t1[i] = (short)(int)t1; L.c_t2[i] = (short)(int)t2;
                    L.c_req[i] = fmin(fmax(x3 * 100.0, 0.0), cap);
                }
            }
            __syncthreads();
            unsigned mine = 0;
            for (int i = lane; i < m; i += 64) {
                const unsigned key = L.c_key[i];
                if (key == 0xffffu) continue;
                mine++;
                int rank = 0;
                for (int j = 0; j < m; j++) rank += L.c_key[j] < key;
                const int pos = have + rank;
                if (pos < want) {
                    L.a_arr[pos] = L.c_t0[i]; L.a_dep[pos] = L.c_t1[i]; L.a_est[pos] = L.c_t2[i];
                    L.a_req[pos] = L.c_req[i];
                }
            }
            have = min(want, have + (int)wave_total_u32(mine));
            __syncthreads();
        }

        // :490 sort by arrival, ties in accept order; EVSE-choice randoms alongside
        const int n = have;
        for (int i = lane; i < n; i += 64) {
            const int key = (int)L.a_arr[i] * 256 + i;
            int rank = 0;
            for (int j = 0; j < n; j++) rank += ((int)L.a_arr[j] * 256 + j) < key;
            L.s_arr[rank] = L.a_arr[i]; L.s_dep[rank] = L.a_dep[i]; L.s_est[rank] = L.a_est[i];
            L.s_req[rank] = L.a_req[i];
            L.rnd[i] = Philox((unsigned)i, 3, e0, e1, k0, k1).w[0];
        }
        __syncthreads();

        // :499-511 lanes = EVSEs
        int station_dep = -1;
        for (int j = 0; j < n; j++) {
            const int a = L.s_arr[j], d = L.s_dep[j];
            const bool avail = lane < n_stations && station_dep < a;
            const int n_avail = __popcll(__ballot(avail));
            int pick = -1;
            if (n_avail > 0) {
                unsigned scan = wave_scan_u32(avail ? usage : 0u);
                unsigned total = (unsigned)__builtin_amdgcn_readlane((int)scan, 63);
                if (total == 0) {                                   // :505-506 uniform among the free EVSEs
                    scan = wave_scan_u32(avail ? 1u : 0u);
                    total = (unsigned)n_avail;
                }
                const unsigned target = (unsigned)(((unsigned long long)L.rnd[j] * (unsigned long long)total) >> 32);
                pick = __builtin_ctzll(__ballot(scan > target));
                if (lane == pick) station_dep = max(d, station_dep);
            }
            if (lane == 0) L.s_station[j] = (short)pick;
        }
        __syncthreads();

        // :514 drop sessions without an EVSE, compact, store (zero padded like evc_upload_episodes)
        int base = 0;
        double profit = 0.0;
        evc_session* out_s = sessions + slot * (size_t)stride;
        double* out_r = requested + slot * (size_t)stride;
        for (int c = 0; c < kGenMaxSessions; c += 64) {
            const int j = c + lane;
            const bool keep = j < n && L.s_station[j] >= 0;
            const unsigned long long mask = __ballot(keep);
            if (keep) {
                const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
                evc_session s;
                s.arrival = L.s_arr[j]; s.departure = L.s_dep[j]; s.est_departure = L.s_est[j];
                s.station = L.s_station[j];
                out_s[pos] = s;
                const double r = L.s_req[j];
                out_r[pos] = r;
                const double cap_kwh = (double)(s.departure - s.arrival) * 32.0 * Consts::A_PERS_TO_KWH;
                profit += fmin(r, cap_kwh) * Consts::MARGINAL_PROFIT_PER_KWH;         // env.py:422-429
            }
            base += __popcll(mask);
        }
        for (int j = base + lane; j < stride; j += 64) {
            evc_session z;
            z.arrival = 0; z.departure = 0; z.est_departure = 0; z.station = 0;
            out_s[j] = z;
            out_r[j] = 0.0;
        }
        profit = wave_sum_f64(profit);
        if (lane == 0) {
            n_sessions[slot] = base;
            slot_moer[slot] = day;
            max_profit[slot] = profit;
        }
        __syncthreads();
    }
}

}  // namespace evc
