// bat_engine.hip — gfx950 implementation of include/battery_dispatch.h (a synthetic workload: see the
// header).  One wavefront per environment: the few scalars of the step are computed wave-uniformly,
// the 4k+6-float observation row is written lane-parallel.  HBM-bound on the observation write
// (600 B per env-step at k = 36) plus the 2k-float bid read.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "evc_hostcopy.h"
using evc::copy_d2h;
using evc::copy_h2d;

#include "../../include/battery_dispatch.h"

namespace {

thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(x)                                                                       \
    do {                                                                                 \
        hipError_t err_ = (x);                                                           \
        if (err_ != hipSuccess) return fail(-4, "%s: %s", #x, hipGetErrorString(err_)); \
    } while (0)

struct BatParams {
    int N, k, F, bank_slots;
    double cap, step_mwh, eta_c, eta_d, e0, pco2;
    double* energy;          // [N]
    double* ret;             // [N] sum of rewards since reset
    int* t;                  // [N]
    int* slot;               // [N]
    const float* price;      // [slots][289]
    const float* load;
    const float* moer;
    const float* load_fc;    // [slots][289 + k]
    const float* moer_fc;
    const double* terminal_price;   // [slots]
};

__device__ __forceinline__ void write_obs(const BatParams& P, float* row, int lane, int t, double e, int slot,
                                          const float* bids, float x, float p, float l, float m) {
    const int k = P.k;
    // [t, e, a(2k), x, p, l, lhat(k), m, mhat(k)]
    for (int i = lane; i < P.F; i += 64) {
        float v;
        if (i == 0) v = (float)t;
        else if (i == 1) v = (float)e;
        else if (i < 2 + 2 * k) v = bids ? bids[i - 2] : 0.0f;
        else if (i == 2 + 2 * k) v = x;
        else if (i == 3 + 2 * k) v = p;
        else if (i == 4 + 2 * k) v = l;
        else if (i < 5 + 3 * k) v = P.load_fc[(size_t)slot * (BAT_TRACE_LEN + k) + t + 1 + (i - (5 + 2 * k))];
        else if (i == 5 + 3 * k) v = m;
        else v = P.moer_fc[(size_t)slot * (BAT_TRACE_LEN + k) + t + 1 + (i - (6 + 3 * k))];
        row[i] = v;
    }
}

__global__ __launch_bounds__(256) void bat_reset_kernel(BatParams P, const int* slots, float* obs) {
    const int lane = threadIdx.x & 63;
    const int env = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (env >= P.N) return;
    const int slot = slots ? slots[env] : env % P.bank_slots;
    if (lane == 0) { P.energy[env] = P.e0; P.ret[env] = 0.0; P.t[env] = 0; P.slot[env] = slot; }
    // first observation: nothing has happened yet (previous action / dispatch / price / load / MOER = 0)
    write_obs(P, obs + (size_t)env * P.F, lane, 0, P.e0, slot, nullptr, 0.0f, 0.0f, 0.0f, 0.0f);
    // forecasts in the first observation start at index 1 like every later one (lhat[t+1 .. t+k])
}

// One step.  Round 3 form: FOUR environments per wavefront (a 16-lane row each), so a 256-thread workgroup steps 16 and a
// launch of 16 384 environments is 1 024 workgroups instead of 4 096; and two dependent memory levels instead of three:
// everything that does not depend on (t, slot) — state, the two bids that decide, the bid row the observation copies — is
// requested at once, then the traces at (slot, t).  The round-2 kernel (one wavefront per environment, t -> slot -> trace
// behind an early-exit branch) ran 14.5 us for 15 MB.
__global__ __launch_bounds__(256) void bat_step_kernel(BatParams P, const float* __restrict__ bids, float* __restrict__ obs,
                                                       double* __restrict__ reward, unsigned char* __restrict__ terminated) {
    const int q = threadIdx.x & 15;
    const int env = (blockIdx.x * 256 + threadIdx.x) >> 4;
    if (env >= P.N) return;
    const int k = P.k, F = P.F;
    // level 1 (no dependencies)
    const int t = P.t[env];
    const int slot = P.slot[env];
    const double e = P.energy[env];
    const double ret_in = P.ret[env];
    const float* a = bids + (size_t)env * 2 * k;
    const float bc = a[0], bd = a[k];
    // The observation row as 8-byte PAIRS (F = 4k + 6 is even, rows are 8-byte aligned): pair 0 = (t, e), pairs 1..k = the bid
    // row, pair k+1 = (x, p), then (l, lhat..., m, mhat...) wherever the pair boundaries fall.  Lane q of the row owns pairs q, q + 16, ...:
    // half the store instructions of a float-per-lane copy.
    constexpr int kPairPasses = (4 * BAT_MAX_FORECAST + 6 + 31) / 32;
    float2 acopy[kPairPasses];
#pragma unroll
    for (int j = 0; j < kPairPasses; j++) {
        const int pr = q + 16 * j;                                  // pair index
        acopy[j] = (pr >= 1 && pr <= k) ? *reinterpret_cast<const float2*>(a + 2 * (pr - 1)) : make_float2(0.0f, 0.0f);
    }
    float* row = obs + (size_t)env * F;
    if (t >= BAT_EPISODE_STEPS) {                       // step after termination: no-op, reward 0
        if (q == 0) { reward[env] = 0.0; terminated[env] = 1; }
        return;
    }
    // level 2 (needs t, slot)
    const size_t tr = (size_t)slot * BAT_TRACE_LEN + t;
    const float pf = P.price[tr], lf = P.load[tr], mf = P.moer[tr];
    const float* lfc = P.load_fc + (size_t)slot * (BAT_TRACE_LEN + k) + t + 2;      // lhat of the NEXT observation: t1 + 1 + j
    const float* mfc = P.moer_fc + (size_t)slot * (BAT_TRACE_LEN + k) + t + 2;
    // forecast floats of the pairs: sources are only 4-byte aligned (they start at t + 2): one float load per element
    float2 fcopy[kPairPasses];
    auto forecast = [&](int i) -> float {                           // output float i, if it is a forecast (one load, source by address)
        const bool is_l = i >= 5 + 2 * k && i < 5 + 3 * k, is_m = i >= 6 + 3 * k && i < 6 + 4 * k;
        const float* src = is_l ? lfc + (i - (5 + 2 * k)) : mfc + (i - (6 + 3 * k));
        return (is_l || is_m) ? *src : 0.0f;
    };
#pragma unroll
    for (int j = 0; j < kPairPasses; j++) {
        const int pr = q + 16 * j;
        fcopy[j] = make_float2(forecast(2 * pr), forecast(2 * pr + 1));
    }
    const int t1 = t + 1;
    const bool done = t1 >= BAT_EPISODE_STEPS;
    const double term_price = done ? P.terminal_price[slot] : 0.0;

    const double p = (double)pf, m = (double)mf;
    const bool sell = p >= (double)bd, buy = p <= (double)bc;
    double x = 0.0, e1 = e;
    if (sell && !buy) {
        x = fmin(P.step_mwh, P.eta_d * e);
        e1 = e - x / P.eta_d;
    } else if (buy && !sell) {
        x = -fmin(P.step_mwh, (P.cap - e) / P.eta_c);
        e1 = e - P.eta_c * x;
    }
    e1 = fmin(fmax(e1, 0.0), P.cap);
    double r = p * x + P.pco2 * m * x;
    if (done) r -= term_price * fmax(0.0, P.e0 - e1);

    // observation [t, e, a(2k), x, p, l, lhat(k), m, mhat(k)]
    float2* row2 = reinterpret_cast<float2*>(row);
    auto scalar = [&](int i, float v) -> float {                    // output float i, if it is one of the six scalars
        if (i == 0) return (float)t1;
        if (i == 1) return (float)e1;
        if (i == 2 + 2 * k) return (float)x;
        if (i == 3 + 2 * k) return pf;
        if (i == 4 + 2 * k) return lf;
        if (i == 5 + 3 * k) return mf;
        return v;
    };
#pragma unroll
    for (int j = 0; j < kPairPasses; j++) {
        const int pr = q + 16 * j;
        float2 v = (pr >= 1 && pr <= k) ? acopy[j] : fcopy[j];
        v.x = scalar(2 * pr, v.x);
        v.y = scalar(2 * pr + 1, v.y);
        if (pr < F / 2) row2[pr] = v;
    }
    if (q == 0) {
        P.energy[env] = e1;
        P.t[env] = t1;
        P.ret[env] = ret_in + r;
        reward[env] = r;
        terminated[env] = done ? 1 : 0;
    }
}

// T steps per launch (bat_rollout): the step above in a loop, state in registers.  Same geometry (a 16-lane row per
// environment, the observation row as 8-byte pairs).  Everything step i + 1 reads — its bid row, the three trace values, the
// forecast floats of its observation — is requested one step AHEAD, before the stores of step i are issued, so that a row always
// has one step's loads in flight behind the stores of the step before: with a trajectory buffer the launch is a stream of 2k-float
// reads and (4k+6)-float writes per environment-step and nothing else.
// Round 5: every memory operation of the loop is a RAW BUFFER access issued unconditionally, predicates folded into the offset
// (kBatOob = dropped by the hardware range check).  The round-4 form used pointer loads / stores behind exec-mask branches
// (`if (pr < F / 2) row[pr] = v`, two conditional loads into one register for the forecast floats): the compiler cannot count
// memory operations across such branches, so every consumer of a loaded value waited with s_waitcnt vmcnt(0) — i.e. for the
// STORES of the previous step as well (vmcnt retires in order): six full drains per step, 5.5 us per step at 0.34 of HBM peak.
// LPE = lanes per environment: 16 (the step kernel's geometry: 4 096 wavefronts for 16 384 environments, 4 per SIMD) or 64.
typedef __amdgpu_buffer_rsrc_t bat_rsrc_t;
typedef unsigned bat_v2u __attribute__((ext_vector_type(2)));
constexpr unsigned kBatOob = 0xfffffff0u;
__device__ __forceinline__ bat_rsrc_t bat_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(unsigned)(bytes > 0xffffff00ull ? 0xffffff00ull : bytes), 0x00020000);
}
__device__ __forceinline__ float bat_ld_f32(bat_rsrc_t r, unsigned off) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0)); }
__device__ __forceinline__ float2 bat_ld_f32x2(bat_rsrc_t r, unsigned off) {
    const bat_v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
template <int AUX = 0>
__device__ __forceinline__ void bat_st_f32x2(bat_rsrc_t r, unsigned off, float2 x) {
    bat_v2u v;
    v.x = __float_as_uint(x.x); v.y = __float_as_uint(x.y);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)off, 0, AUX);
}
__device__ __forceinline__ void bat_st_f64(bat_rsrc_t r, unsigned off, double x) {
    bat_v2u v;
    v.x = (unsigned)__double2loint(x); v.y = (unsigned)__double2hiint(x);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)off, 0, 0);
}

// TRAJ: with a trajectory buffer (every step's observation row is stored: bid pairs and forecast floats are loaded every step);
// without one a step reads five scalars and stores nothing but the optional reward.  The LAST live step's observation row (the
// persistent `obs` output) is rebuilt after the loop from what the loop kept of that step, in both forms.
// NT: trajectory rows start on 128-byte lines and are a whole number of lines long (pitch % 32 == 0, aligned base): their stores
// carry the non-temporal hint (written once, read by somebody else much later): 3.77 -> 3.28 us per step.  Packed 600-byte rows
// straddle lines and must merge in L2: with nt they lose (3.9 -> 5.4), so they keep the default policy.
// KC: the forecast length as a compile-time constant (0: P.k at run time).  The loop is bound by instruction issue, not by memory (a fill
// kernel writes 6.9 TB/s on this GPU, tools/probes/hbm_rw_mix.py): with a run-time k every float of a row costs range compares
// against 2k+2, 3k+5, ... and a six-way select for the scalars of the observation; with k = 36 (the reference's forecast horizon,
// the default) the compiler folds them per unrolled pass — most passes hold no scalar at all.
template <int LPE, bool TRAJ, bool NT, int KC>
__global__ __launch_bounds__(256) void bat_rollout_kernel(BatParams P, const float* __restrict__ ring, int ring_len, int steps,
                                                          float* __restrict__ obs, double* __restrict__ reward,
                                                          unsigned char* __restrict__ terminated, float* __restrict__ obs_traj,
                                                          int traj_pitch, double* __restrict__ reward_traj) {
    const int q = threadIdx.x & (LPE - 1);
    const int env_i = (blockIdx.x * 256 + threadIdx.x) / LPE;
    const bool ev = env_i < P.N;                       // rows past N issue the same accesses, all out of range
    const unsigned env = ev ? (unsigned)env_i : 0u;
    const int k = KC > 0 ? KC : P.k, F = KC > 0 ? 4 * KC + 6 : P.F;
    const unsigned N = (unsigned)P.N;
    constexpr int kPairPasses = ((4 * BAT_MAX_FORECAST + 6) / 2 + LPE - 1) / LPE;
    int t = ev ? P.t[env] : BAT_EPISODE_STEPS;
    const unsigned slot = ev ? (unsigned)P.slot[env] : 0u;
    double e = ev ? P.energy[env] : 0.0;
    double ret = ev ? P.ret[env] : 0.0;
    const double term_price = P.terminal_price[slot];
    const unsigned S = (unsigned)P.bank_slots, Tk = (unsigned)(BAT_TRACE_LEN + k);
    const bat_rsrc_t r_ring = bat_rsrc(ring, (size_t)ring_len * N * 2u * k * 4u);
    const bat_rsrc_t r_price = bat_rsrc(P.price, (size_t)S * BAT_TRACE_LEN * 4u), r_load = bat_rsrc(P.load, (size_t)S * BAT_TRACE_LEN * 4u),
                     r_moer = bat_rsrc(P.moer, (size_t)S * BAT_TRACE_LEN * 4u);
    // the two forecast tables are ONE allocation (bat_create): [load_fc | moer_fc], so that a lane picks its table by offset
    const bat_rsrc_t r_fc = bat_rsrc(P.load_fc, (size_t)2u * S * Tk * 4u);
    const bat_rsrc_t r_obs = bat_rsrc(obs, (size_t)N * F * 4u);
    const bat_rsrc_t r_rt = bat_rsrc(reward_traj, reward_traj ? (size_t)steps * N * 8u : 0u);
    const unsigned row_bids = 2u * (unsigned)k * 4u, batch_bids = N * row_bids;
    struct StepIn { float2 a[TRAJ ? kPairPasses : 1], f[TRAJ ? kPairPasses : 1]; float bc, bd, pf, lf, mf; };
    // the bid pairs and forecast floats of the observation row after step i (taken at period tt)
    auto fetch_row = [&](int i, int tt, bool on, float2 (&a)[kPairPasses], float2 (&f)[kPairPasses]) {
        const unsigned base = (unsigned)(i % ring_len) * batch_bids + env * row_bids;
        const unsigned tc = (unsigned)(tt < BAT_EPISODE_STEPS ? tt : BAT_EPISODE_STEPS - 1);
        const unsigned fc0 = slot * Tk + tc + 2u;              // lhat / mhat of the NEXT observation start at t1 + 1
#pragma unroll
        for (int j = 0; j < kPairPasses; j++) {
            const int pr = q + LPE * j;
            a[j] = bat_ld_f32x2(r_ring, (on && pr >= 1 && pr <= k) ? base + 8u * (unsigned)(pr - 1) : kBatOob);
            bool in[2];
            unsigned idx[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int ii = 2 * pr + h;
                const bool is_l = ii >= 5 + 2 * k && ii < 5 + 3 * k, is_m = ii >= 6 + 3 * k && ii < 6 + 4 * k;
                idx[h] = is_l ? fc0 + (unsigned)(ii - (5 + 2 * k)) : S * Tk + fc0 + (unsigned)(ii - (6 + 3 * k));
                in[h] = on && (is_l || is_m);
            }
            if constexpr (KC > 0) {
                // one 8-byte load per pair: the two floats of a pair are neighbours in their table, and where only one of them is a
                // table value (the pairs that also hold lf / mf) the other half reads the neighbouring table entry and is replaced by
                // the scalar in row_value (buffer loads need 4-byte alignment only)
                f[j] = bat_ld_f32x2(r_fc, in[0] ? idx[0] * 4u : (in[1] ? (idx[1] - 1u) * 4u : kBatOob));
            } else {
                f[j] = make_float2(bat_ld_f32(r_fc, in[0] ? idx[0] * 4u : kBatOob), bat_ld_f32(r_fc, in[1] ? idx[1] * 4u : kBatOob));
            }
        }
    };
    // everything step i reads, for the environment's period tt; `on` = the step exists and the row is still running
    auto fetch = [&](int i, int tt, bool on, StepIn& in) {
        const unsigned base = (unsigned)(i % ring_len) * batch_bids + env * row_bids;
        in.bc = bat_ld_f32(r_ring, on ? base : kBatOob);
        in.bd = bat_ld_f32(r_ring, on ? base + (unsigned)k * 4u : kBatOob);
        const unsigned tc = (unsigned)(tt < BAT_EPISODE_STEPS ? tt : BAT_EPISODE_STEPS - 1);
        const unsigned tr = on ? (slot * BAT_TRACE_LEN + tc) * 4u : kBatOob;
        in.pf = bat_ld_f32(r_price, tr); in.lf = bat_ld_f32(r_load, tr); in.mf = bat_ld_f32(r_moer, tr);
        if constexpr (TRAJ) fetch_row(i, tt, on, in.a, in.f);
    };
    auto row_value = [&](int pr, float2 a, float2 f, int t1, double e1, double x, float pf, float lf, float mf) {
        float2 v = (pr >= 1 && pr <= k) ? a : f;
        auto scalar = [&](int ii, float w) -> float {
            if (ii == 0) return (float)t1;
            if (ii == 1) return (float)e1;
            if (ii == 2 + 2 * k) return (float)x;
            if (ii == 3 + 2 * k) return pf;
            if (ii == 4 + 2 * k) return lf;
            if (ii == 5 + 3 * k) return mf;
            return w;
        };
        v.x = scalar(2 * pr, v.x);
        v.y = scalar(2 * pr + 1, v.y);
        return pr < F / 2 ? v : make_float2(0.0f, 0.0f);
    };
    StepIn cur, nxt;
    fetch(0, t, t < BAT_EPISODE_STEPS, cur);
    double r = 0.0;
    bool done = t >= BAT_EPISODE_STEPS;
    // the last live step, for the `obs` row: its index and the scalars of its observation (t and e are the final state)
    int i_last = -1;
    double x_last = 0.0;
    float pf_last = 0.0f, lf_last = 0.0f, mf_last = 0.0f;
    for (int i = 0; i < steps; i++) {
        const bool act = t < BAT_EPISODE_STEPS;          // steps after termination: no-ops, reward 0 (rows of obs_traj not written)
        const int t1 = t + 1;
        const bool done1 = t1 >= BAT_EPISODE_STEPS;
        fetch(i + 1, t1, act && !done1 && i + 1 < steps, nxt);
        const float pf = cur.pf, lf = cur.lf, mf = cur.mf;
        const double p = (double)pf, m = (double)mf;
        const bool sell = p >= (double)cur.bd, buy = p <= (double)cur.bc;
        double x = 0.0, e1 = e;
        if (sell && !buy) {
            x = fmin(P.step_mwh, P.eta_d * e);
            e1 = e - x / P.eta_d;
        } else if (buy && !sell) {
            x = -fmin(P.step_mwh, (P.cap - e) / P.eta_c);
            e1 = e - P.eta_c * x;
        }
        e1 = fmin(fmax(e1, 0.0), P.cap);
        double ri = p * x + P.pco2 * m * x;
        if (done1) ri -= term_price * fmax(0.0, P.e0 - e1);
        if constexpr (TRAJ) {
            // trajectory rows at the caller's pitch (floats): the descriptor of step i's slab [N][pitch].  NT (whole-line form,
            // the caller said the padding is the kernel's: bat_rollout_pitched with a negative pitch): the padding behind the
            // 4k+6 floats is written too (zeros) so that whole lines go out; otherwise nothing behind the row is touched
            const bat_rsrc_t r_traj = bat_rsrc(obs_traj + (size_t)i * N * (size_t)traj_pitch, (size_t)N * (size_t)traj_pitch * 4u);
#pragma unroll
            for (int j = 0; j < kPairPasses; j++) {
                const int pr = q + LPE * j;
                const float2 v = row_value(pr, cur.a[j], cur.f[j], t1, e1, x, pf, lf, mf);
                bat_st_f32x2<NT ? 2 : 0>(r_traj, (act && ev && pr < (NT ? traj_pitch : F) / 2) ? (env * (unsigned)traj_pitch + 2u * (unsigned)pr) * 4u : kBatOob, v);
            }
        }
        bat_st_f64(r_rt, (ev && q == 0) ? ((unsigned)i * N + env) * 8u : kBatOob, act ? ri : 0.0);
        if (act) {
            r = ri;
            ret += ri;
            e = e1;
            t = t1;
            done = done1;
            i_last = i; x_last = x; pf_last = pf; lf_last = lf; mf_last = mf;
        } else {
            r = 0.0;
            done = true;
        }
        cur = nxt;
    }
    {   // `obs` row of the last live step (none: the environment had ended before the call; its row stays as it was)
        const bool on = ev && i_last >= 0;
        float2 a[kPairPasses], f[kPairPasses];
        fetch_row(i_last < 0 ? 0 : i_last, t - 1, on, a, f);
#pragma unroll
        for (int j = 0; j < kPairPasses; j++) {
            const int pr = q + LPE * j;
            const float2 v = row_value(pr, a[j], f[j], t, e, x_last, pf_last, lf_last, mf_last);
            bat_st_f32x2(r_obs, (on && pr < F / 2) ? (env * (unsigned)F + 2u * (unsigned)pr) * 4u : kBatOob, v);
        }
    }
    if (q == 0 && ev) {
        P.energy[env] = e;
        P.t[env] = t;
        P.ret[env] = ret;
        reward[env] = r;
        terminated[env] = done ? 1 : 0;
    }
}

__global__ void bat_metrics_kernel(BatParams P, double* out) {
    double se = 0.0, sr = 0.0, done = 0.0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < P.N; e += gridDim.x * blockDim.x) {
        se += P.energy[e]; sr += P.ret[e]; done += P.t[e] >= BAT_EPISODE_STEPS ? 1.0 : 0.0;
    }
    atomicAdd(&out[0], se); atomicAdd(&out[1], sr); atomicAdd(&out[3], done);
}

}  // namespace

struct bat_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    BatParams P{};
    bat_config cfg{};
    unsigned long long env_steps = 0;
    int* d_slots = nullptr;
    float* d_bids = nullptr;
    float* d_obs = nullptr;
    double* d_reward = nullptr;
    unsigned char* d_term = nullptr;
    double* d_metrics = nullptr;
    std::vector<void*> owned;
};

extern "C" {

const char* bat_last_error(void) { return g_err; }

int bat_create(const bat_config* cfg, bat_engine** out) {
    if (!cfg || !out) return fail(-1, "bat_create: null argument");
    if (cfg->num_envs < 1 || cfg->forecast_steps < 1 || cfg->forecast_steps > BAT_MAX_FORECAST || cfg->bank_slots < 1)
        return fail(-1, "bat_create: bad num_envs / forecast_steps / bank_slots");
    if (!(cfg->capacity_mwh > 0) || !(cfg->max_power_mw > 0) || !(cfg->eta_charge > 0 && cfg->eta_charge <= 1) ||
        !(cfg->eta_discharge > 0 && cfg->eta_discharge <= 1) || !(cfg->init_energy_mwh >= 0 && cfg->init_energy_mwh <= cfg->capacity_mwh))
        return fail(-1, "bat_create: bad battery parameters");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= cfg->device)
        return fail(-2, "bat_create: no HIP device %d (there is no CPU path)", cfg->device);
    HIP_TRY(hipSetDevice(cfg->device));
    bat_engine* e = new bat_engine();
    e->device = cfg->device;
    e->cfg = *cfg;
    BatParams& P = e->P;
    P.N = cfg->num_envs; P.k = cfg->forecast_steps; P.F = 4 * P.k + 6; P.bank_slots = cfg->bank_slots;
    P.cap = cfg->capacity_mwh; P.step_mwh = cfg->max_power_mw * (5.0 / 60.0);
    P.eta_c = cfg->eta_charge; P.eta_d = cfg->eta_discharge; P.e0 = cfg->init_energy_mwh; P.pco2 = cfg->co2_price_per_kg;
    const size_t N = P.N, S = P.bank_slots, k = P.k;
    auto alloc = [&](void** p, size_t bytes) -> hipError_t {
        hipError_t r = hipMalloc(p, bytes);
        if (r == hipSuccess) { e->owned.push_back(*p); r = hipMemset(*p, 0, bytes); }
        return r;
    };
#define BA(ptr, bytes) do { if (alloc((void**)&(ptr), (bytes)) != hipSuccess) { bat_destroy(e); return fail(-3, "bat_create: hipMalloc failed"); } } while (0)
    BA(P.energy, N * 8); BA(P.ret, N * 8); BA(P.t, N * 4); BA(P.slot, N * 4);
    BA(P.price, S * BAT_TRACE_LEN * 4); BA(P.load, S * BAT_TRACE_LEN * 4); BA(P.moer, S * BAT_TRACE_LEN * 4);
    BA(P.load_fc, 2 * S * (BAT_TRACE_LEN + k) * 4);      // [load_fc | moer_fc]: one allocation (bat_rollout_kernel picks the table by offset)
    P.moer_fc = P.load_fc + S * (BAT_TRACE_LEN + k);
    BA(P.terminal_price, S * 8);
    BA(e->d_slots, N * 4); BA(e->d_bids, N * 2 * k * 4); BA(e->d_obs, N * (4 * k + 6) * 4); BA(e->d_reward, N * 8);
    BA(e->d_term, N); BA(e->d_metrics, 4 * 8);
#undef BA
    *out = e;
    return 0;
}

void bat_destroy(bat_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    for (void* p : e->owned) (void)hipFree(p);
    delete e;
}

int bat_obs_dim(const bat_engine* e) { return e ? e->P.F : -1; }
int bat_set_stream(bat_engine* e, void* s) { if (!e) return fail(-1, "null engine"); e->stream = (hipStream_t)s; return 0; }

int bat_upload_traces(bat_engine* e, int32_t first, int32_t count, const float* price, const float* load,
                      const float* load_fc, const float* moer, const float* moer_fc, const double* terminal_price) {
    if (!e || !price || !load || !load_fc || !moer || !moer_fc || !terminal_price) return fail(-1, "bat_upload_traces: null argument");
    if (first < 0 || count < 1 || first + count > e->P.bank_slots) return fail(-1, "bat_upload_traces: slots outside the bank");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const size_t T = BAT_TRACE_LEN, Tk = BAT_TRACE_LEN + e->P.k, c = count, f = first;
    HIP_TRY(copy_h2d((void*)(e->P.price + f * T), price, c * T * 4, e->stream));
    HIP_TRY(copy_h2d((void*)(e->P.load + f * T), load, c * T * 4, e->stream));
    HIP_TRY(copy_h2d((void*)(e->P.moer + f * T), moer, c * T * 4, e->stream));
    HIP_TRY(copy_h2d((void*)(e->P.load_fc + f * Tk), load_fc, c * Tk * 4, e->stream));
    HIP_TRY(copy_h2d((void*)(e->P.moer_fc + f * Tk), moer_fc, c * Tk * 4, e->stream));
    HIP_TRY(copy_h2d((void*)(e->P.terminal_price + f), terminal_price, c * 8, e->stream));
    return 0;
}

int bat_reset(bat_engine* e, const int32_t* slots, float* obs_dev) {
    if (!e || !obs_dev) return fail(-1, "bat_reset: null argument");
    HIP_TRY(hipSetDevice(e->device));
    const int* dslots = nullptr;
    if (slots) {
        for (int i = 0; i < e->P.N; i++)
            if (slots[i] < 0 || slots[i] >= e->P.bank_slots) return fail(-1, "bat_reset: slot %d outside the bank", slots[i]);
        HIP_TRY(copy_h2d(e->d_slots, slots, sizeof(int) * e->P.N, e->stream));
        dslots = e->d_slots;
    }
    hipLaunchKernelGGL(bat_reset_kernel, dim3((e->P.N + 3) / 4), dim3(256), 0, e->stream, e->P, dslots, obs_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int bat_step(bat_engine* e, const float* bids_dev, float* obs_dev, double* reward_dev, uint8_t* terminated_dev) {
    if (!e || !bids_dev || !obs_dev || !reward_dev || !terminated_dev) return fail(-1, "bat_step: null argument");
    HIP_TRY(hipSetDevice(e->device));
    hipLaunchKernelGGL(bat_step_kernel, dim3((e->P.N + 15) / 16), dim3(256), 0, e->stream, e->P, bids_dev, obs_dev,
                       reward_dev, terminated_dev);
    HIP_TRY(hipGetLastError());
    e->env_steps += (unsigned long long)e->P.N;
    return 0;
}

int bat_rollout(bat_engine* e, const float* bids_ring_dev, int32_t ring_len, int32_t steps, float* obs_dev, double* reward_dev,
                uint8_t* terminated_dev, float* obs_traj_dev, double* reward_traj_dev) {
    return bat_rollout_pitched(e, bids_ring_dev, ring_len, steps, obs_dev, reward_dev, terminated_dev, obs_traj_dev,
                               e ? e->P.F : 0, reward_traj_dev);
}

int bat_rollout_pitched(bat_engine* e, const float* bids_ring_dev, int32_t ring_len, int32_t steps, float* obs_dev, double* reward_dev,
                        uint8_t* terminated_dev, float* obs_traj_dev, int32_t traj_pitch, double* reward_traj_dev) {
    if (!e || !bids_ring_dev || !obs_dev || !reward_dev || !terminated_dev) return fail(-1, "bat_rollout: null argument");
    if (ring_len < 1 || steps < 1) return fail(-1, "bat_rollout: ring_len and steps must be >= 1");
    // a NEGATIVE pitch: rows |pitch| floats apart AND the floats behind each row (up to the pitch) are the kernel's to fill with
    // zeros — whole 128-byte lines go out with non-temporal stores.  A positive pitch never writes behind the 4k+6 floats: the
    // caller may keep other columns there (`out[0]` as a [:, :, :F] slice of a wider tensor).
    const bool pad_mine = traj_pitch < 0;
    if (pad_mine) traj_pitch = -traj_pitch;
    if (obs_traj_dev && (traj_pitch < e->P.F || (traj_pitch & 1)))
        return fail(-1, "bat_rollout: trajectory pitch %d must be even and >= the observation width %d", traj_pitch, e->P.F);
    // the kernel addresses the bid ring, a step's trajectory slab and the reward trajectory through 32-bit buffer offsets
    // (raw buffer descriptors: num_records < 4 GiB); larger ones would wrap or be range-dropped silently
    {
        const unsigned long long lim = 0xffffff00ull, N = (unsigned long long)e->P.N;
        if ((unsigned long long)ring_len * N * 2ull * (unsigned long long)e->P.k * 4ull > lim)
            return fail(-1, "bat_rollout: bid ring of %d x %d x %d floats exceeds the 4 GiB a launch can address: pass a shorter ring "
                            "(ring_len <= %llu) or roll out in chunks", ring_len, e->P.N, 2 * e->P.k, lim / (N * 2ull * (unsigned long long)e->P.k * 4ull));
        if (reward_traj_dev && (unsigned long long)steps * N * 8ull > lim)
            return fail(-1, "bat_rollout: reward trajectory of %d x %d doubles exceeds 4 GiB: roll out in chunks of <= %llu steps", steps, e->P.N, lim / (N * 8ull));
        if (obs_traj_dev && N * (unsigned long long)traj_pitch * 4ull > lim)
            return fail(-1, "bat_rollout: one step's trajectory slab (%d rows of %d floats) exceeds 4 GiB", e->P.N, traj_pitch);
    }
    HIP_TRY(hipSetDevice(e->device));
    // lanes per environment: 16 = the step kernel's geometry (default); 64 = a wavefront per environment, measured SLOWER
    // (7.2 against 5.5 us per 16 384-environment step with a trajectory, 3.0 against 1.6 without): the loop is not short of
    // loops in flight.  BAT_ROLLOUT_LPE=64 selects it (measurements).
    static const int lpe = getenv("BAT_ROLLOUT_LPE") ? atoi(getenv("BAT_ROLLOUT_LPE")) : 16;
    if (lpe != 16 && lpe != 64) return fail(-1, "BAT_ROLLOUT_LPE=%d: 16 or 64", lpe);
#define BAT_LAUNCH_ROLLOUT_(L, T, NTF, KC, GRID)                                                                               \
    hipLaunchKernelGGL((bat_rollout_kernel<L, T, NTF, KC>), dim3(GRID), dim3(256), 0, e->stream, e->P, bids_ring_dev, ring_len, steps, obs_dev, \
                       reward_dev, terminated_dev, obs_traj_dev, traj_pitch, reward_traj_dev)
    // (k = 36 compiled in — the default horizon; any other k: the run-time form.  BAT_ROLLOUT_KC=0: measurements)
    static const bool kc_ok = !(getenv("BAT_ROLLOUT_KC") && atoi(getenv("BAT_ROLLOUT_KC")) == 0);
    const bool k36 = kc_ok && e->P.k == 36 && e->P.F == 4 * 36 + 6;
#define BAT_LAUNCH_ROLLOUT(L, T, NTF, GRID)                                                                                    \
    do { if (k36) BAT_LAUNCH_ROLLOUT_(L, T, NTF, 36, GRID); else BAT_LAUNCH_ROLLOUT_(L, T, NTF, 0, GRID); } while (0)
    const bool traj = obs_traj_dev != nullptr;
    const bool lines = traj && pad_mine && traj_pitch % 32 == 0 && ((uintptr_t)obs_traj_dev & 127u) == 0;
    const int g16 = (e->P.N + 15) / 16, g64 = (e->P.N + 3) / 4;
    if (lpe == 16) {
        if (!traj) BAT_LAUNCH_ROLLOUT(16, false, false, g16);
        else if (lines) BAT_LAUNCH_ROLLOUT(16, true, true, g16);
        else BAT_LAUNCH_ROLLOUT(16, true, false, g16);
    } else {
        if (!traj) BAT_LAUNCH_ROLLOUT(64, false, false, g64);
        else if (lines) BAT_LAUNCH_ROLLOUT(64, true, true, g64);
        else BAT_LAUNCH_ROLLOUT(64, true, false, g64);
    }
#undef BAT_LAUNCH_ROLLOUT
#undef BAT_LAUNCH_ROLLOUT_
    HIP_TRY(hipGetLastError());
    e->env_steps += (unsigned long long)e->P.N * (unsigned long long)steps;
    return 0;
}

int bat_reset_host(bat_engine* e, const int32_t* slots, float* obs_host) {
    if (!e || !obs_host) return fail(-1, "bat_reset_host: null argument");
    if (int rc = bat_reset(e, slots, e->d_obs)) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(copy_d2h(obs_host, e->d_obs, sizeof(float) * (size_t)e->P.N * e->P.F, e->stream));
    return 0;
}

int bat_step_host(bat_engine* e, const float* bids_host, float* obs_host, double* reward_host, uint8_t* terminated_host) {
    if (!e || !bids_host || !obs_host || !reward_host || !terminated_host) return fail(-1, "bat_step_host: null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(copy_h2d(e->d_bids, bids_host, sizeof(float) * (size_t)e->P.N * 2 * e->P.k, e->stream));
    if (int rc = bat_step(e, e->d_bids, e->d_obs, e->d_reward, e->d_term)) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(copy_d2h(obs_host, e->d_obs, sizeof(float) * (size_t)e->P.N * e->P.F, e->stream));
    HIP_TRY(copy_d2h(reward_host, e->d_reward, sizeof(double) * e->P.N, e->stream));
    HIP_TRY(copy_d2h(terminated_host, e->d_term, e->P.N, e->stream));
    return 0;
}

int bat_get_state(bat_engine* e, double* energy_host, int32_t* t_host) {
    if (!e) return fail(-1, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (energy_host) HIP_TRY(copy_d2h(energy_host, e->P.energy, sizeof(double) * e->P.N, e->stream));
    if (t_host) HIP_TRY(copy_d2h(t_host, e->P.t, sizeof(int) * e->P.N, e->stream));
    return 0;
}

int bat_read_metrics(bat_engine* e, double* out_host) {
    if (!e || !out_host) return fail(-1, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemsetAsync(e->d_metrics, 0, 4 * sizeof(double), e->stream));
    hipLaunchKernelGGL(bat_metrics_kernel, dim3(64), dim3(256), 0, e->stream, e->P, e->d_metrics);
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(copy_d2h(out_host, e->d_metrics, 4 * sizeof(double), e->stream));
    out_host[2] = (double)e->env_steps;
    return 0;
}

}  // extern "C"
