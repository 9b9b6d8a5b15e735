// evc_kernels.h — the step / reset / projection kernels (gfx950, wave-per-environment).
#pragma once

#include "evc_device.h"

namespace evc {

// LDS image of the network tables needed by one workgroup (only the [G][m] corner).
struct LdsNet {
    double Mre[EVC_MAX_GROUPS][EVC_MAX_CONSTRAINTS];
    double Mim[EVC_MAX_GROUPS][EVC_MAX_CONSTRAINTS];
    double Aabs[EVC_MAX_GROUPS][EVC_MAX_CONSTRAINTS];
    double mag[EVC_MAX_CONSTRAINTS];
};

__device__ __forceinline__ void stage_net(LdsNet& s, const Params& P) {
    const int m = P.m, G = P.G;
    for (int idx = threadIdx.x; idx < G * m; idx += blockDim.x) {
        int g = idx / m, c = idx - g * m;
        s.Mre[g][c] = P.tables->Mre[g][c];
        s.Mim[g][c] = P.tables->Mim[g][c];
        s.Aabs[g][c] = P.tables->Aabs[g][c];
    }
    for (int c = threadIdx.x; c < m; c += blockDim.x) s.mag[c] = P.tables->mag[c];
    __syncthreads();
}

// Per-lane / per-wave state of the environment a wave is stepping.
struct EnvRegs {
    // per station (lane)
    double rem;   // remaining demand (kWh)
    int dep;      // departure period, kEmptyDep if the EVSE is empty
    int est;      // estimated departure period
    // per environment (wave-uniform)
    int t, cursor, slot, moer_day, n_sessions, next_arrival, status, episodes;
};

__device__ __forceinline__ void load_env(const Params& P, int env, int lane, EnvRegs& r) {
    const int4 s0 = P.scal[2 * env], s1 = P.scal[2 * env + 1];
    r.t = rfl(s0.x); r.cursor = rfl(s0.y); r.slot = rfl(s0.z); r.moer_day = rfl(s0.w);
    r.n_sessions = rfl(s1.x); r.next_arrival = rfl(s1.y); r.status = rfl(s1.z); r.episodes = rfl(s1.w);
    if (lane < P.n) {
        r.rem = P.rem[(size_t)env * P.n + lane];
        int de = P.depest[(size_t)env * P.n + lane];
        r.dep = (int)(short)(de & 0xffff);
        r.est = de >> 16;
    } else {
        r.rem = 0.0; r.dep = kEmptyDep; r.est = 0;
    }
}

__device__ __forceinline__ void store_env(const Params& P, int env, int lane, const EnvRegs& r) {
    if (lane < P.n) {
        P.rem[(size_t)env * P.n + lane] = r.rem;
        P.depest[(size_t)env * P.n + lane] = (r.dep & 0xffff) | (r.est << 16);
    }
    if (lane == 0) {
        P.scal[2 * env] = make_int4(r.t, r.cursor, r.slot, r.moer_day);
        P.scal[2 * env + 1] = make_int4(r.n_sessions, r.next_arrival, r.status, r.episodes);
    }
}

// Puts environment registers into the state right after EVChargingEnv.reset (env.py:319-333)
// for bank slot `slot`.
__device__ __forceinline__ void reset_regs(const Params& P, int slot, EnvRegs& r) {
    r.rem = 0.0; r.dep = kEmptyDep; r.est = 0;
    r.t = 0; r.cursor = 0; r.slot = slot;
    r.moer_day = rfl(P.slot_moer_day[slot]);
    r.n_sessions = rfl(P.n_sessions[slot]);
    r.next_arrival = (r.n_sessions > 0)
        ? rfl((int)P.sessions[(size_t)slot * P.max_sessions].arrival) : kNoArrival;
}

// env.py:381-394: observation row [demands | est_departures | forecasted_moer | prev_moer | t/288]
__device__ __forceinline__ void write_obs(const Params& P, float* row, int lane, const EnvRegs& r) {
    const int n = P.n, k = P.k;
    const bool active = (r.dep != kEmptyDep) && (r.rem > Consts::FULLY_CHARGED_EPS);
    const size_t mrow = ((size_t)r.moer_day * EVC_MOER_ROWS + r.t) * EVC_MOER_COLS;
    if (lane < n) {
        row[lane] = active ? (float)r.rem : 0.0f;
        row[n + lane] = active ? (float)(r.est - r.t) : 0.0f;
    }
    if (lane < k) row[2 * n + lane] = P.moer_obs[mrow + 1 + lane];
    if (lane == 0) {
        row[2 * n + k] = P.moer_obs[mrow];
        row[2 * n + k + 1] = (float)((double)r.t / (double)EVC_EPISODE_STEPS);
    }
}

// Normalised action of this lane's station, clamped to [0,1] (the reference raises on
// out-of-range actions), as float64.  DiscreteActionWrapper.action
// (wrappers.py:43-45) divides in float32.
__device__ __forceinline__ double load_action(const Params& P, const StepIO& io, int env, int lane,
                                              bool& clamped) {
    double a = 0.0;
    clamped = false;
    if (lane < P.n) {
        size_t idx = (size_t)env * P.n + lane;
        if (io.action_kind == EVC_ACTION_DISCRETE) {
            long long v = ((const long long*)io.actions)[idx];
            a = (double)((float)v / (float)(io.bins - 1));
        } else {
            a = (double)((const float*)io.actions)[idx];
        }
        if (!(a >= 0.0)) { clamped = (a != 0.0); a = 0.0; }
        else if (a > 1.0) { clamped = true; a = 1.0; }
    }
    return a;
}

// Upper bound of the projected action (amps): min(32, demand_f32 / A_PERS_TO_KWH)
// (env.py:188-189 with demands = previous float32 observation, env.py:218).
__device__ __forceinline__ double demand_cap_amps(const EnvRegs& r) {
    const bool active = (r.dep != kEmptyDep) && (r.rem > Consts::FULLY_CHARGED_EPS);
    double demand = active ? (double)(float)r.rem : 0.0;
    double u = demand / Consts::A_PERS_TO_KWH / Consts::ACTION_SCALE_FACTOR;
    if (u > 1.0) u = 1.0;
    return u * Consts::ACTION_SCALE_FACTOR;
}

// ||M_c S|| for row c = lane (lane < m) where the class sums S_g are integers given as
// bit-planes (wave-uniform, computed on the scalar unit).  `scale` rescales the integer sums.
template <int BITS>
__device__ __forceinline__ double row_magnitude_planes(const LdsNet& net, const Params& P,
                                                       int lane_c, const BitPlanes<BITS>& planes) {
    double re = 0.0, im = 0.0;
    for (int g = 0; g < P.G; g++) {
        const double s = (double)planes.sum(P.group_mask[g]);
        re += net.Mre[g][lane_c] * s;
        im += net.Mim[g][lane_c] * s;
    }
    return sqrt(re * re + im * im);
}

// Everything of EVChargingEnv.step after the projection: rounding to legal pilots, one
// acnsim.Simulator.step pass, observation, reward, bookkeeping, autoreset.
//   y      : this lane's (projected) action in amps, before rounding
//   regs   : environment state loaded by load_env
__device__ __forceinline__ void finish_step(const Params& P, const StepIO& io, const LdsNet& net,
                                            int env, int lane, double y, double xproj,
                                            bool clamped, EnvRegs& r) {
    const int n = P.n, m = P.m;
    const bool is_cc = (P.cc_mask >> lane) & 1ull;
    const bool in_net = lane < n;

    // ---- env.py:279 ----
    const int t1 = r.t + 1;

    // ---- env.py:366-378 pilots ----
    double pilot = in_net ? legal_pilot(y, is_cc) : 0.0;

    // ---- acnsim update_pilots: charge the plugged EVs for iteration t1-1 ----
    const bool occupied = r.dep != kEmptyDep;
    double amps = 0.0;
    if (occupied) amps = charge_ev(pilot, r.rem);
    const double total_rate = wave_sum_f64(amps);                   // env.py:445

    // ---- env.py:449-452 constraint violation of the PILOT schedule ----
    BitPlanes<6> planes;                                            // pilots are integers <= 32
    planes.build((int)pilot);
    double excess = 0.0;
    {
        double ex = 0.0;
        if (lane < m) {
            ex = row_magnitude_planes(net, P, lane, planes) - net.mag[lane];
            ex = ex > 0.0 ? ex : 0.0;
        }
        if (__ballot(ex > 0.0) != 0ull) excess = wave_sum_f64(ex);
    }

    // ---- acnsim event pass at iteration t1: unplug (precedence 0) before plug-in (10) ----
    if (occupied && r.dep <= t1) { r.dep = kEmptyDep; r.est = 0; r.rem = 0.0; }
    while (r.next_arrival <= t1 && r.cursor < r.n_sessions) {
        const size_t sidx = (size_t)r.slot * P.max_sessions + r.cursor;
        const evc_session s = P.sessions[sidx];
        const double rq = P.requested[sidx];
        const int st = rfl((int)s.station);
        const bool mine = lane == st;
        const bool busy = mine && (r.dep != kEmptyDep);
        if (__ballot(busy) != 0ull) {
            r.status |= EVC_STATUS_OCCUPIED;          // acnportal: StationOccupiedError
        } else if (mine) {
            r.dep = (int)s.departure; r.est = (int)s.est_departure; r.rem = rq;
        }
        r.cursor += 1;
        r.next_arrival = (r.cursor < r.n_sessions)
            ? rfl((int)P.sessions[sidx + 1].arrival) : kNoArrival;
    }
    r.t = t1;
    if (__ballot(clamped) != 0ull) r.status |= EVC_STATUS_ACTION_CLAMPED;

    // ---- env.py:431-464 reward ----
    const double moer_now = P.moer_hist[(size_t)r.moer_day * EVC_MOER_ROWS + t1];
    const double profit = Consts::PROFIT_FACTOR * total_rate;
    const double carbon = Consts::CARBON_COST_FACTOR * total_rate * moer_now;
    const double excess_charge = excess * Consts::VIOLATION_FACTOR;
    const double reward = profit - carbon - excess_charge;
    double acc = 0.0;
    if (lane < 3) {
        acc = P.acc[(size_t)env * 3 + lane];
        acc += (lane == 0) ? profit : (lane == 1 ? carbon : excess_charge);
    }
    const bool done = t1 >= EVC_EPISODE_STEPS;     // event queue empty after the pass at 288

    // ---- outputs ----
    if (lane == 0) {
        io.out.reward[env] = reward;
        io.out.terminated[env] = done ? 1 : 0;
    }
    if (io.out.breakdown && lane < 3) io.out.breakdown[(size_t)env * 3 + lane] = acc;
    if (in_net) {
        if (io.out.pilots) io.out.pilots[(size_t)env * n + lane] = pilot;
        if (io.out.rates) io.out.rates[(size_t)env * n + lane] = amps;
        if (io.out.projected) io.out.projected[(size_t)env * n + lane] = xproj;
    }
    float* obs_row = io.out.obs + (size_t)env * P.F;
    if (done && P.autoreset) {
        if (io.out.final_obs) write_obs(P, io.out.final_obs + (size_t)env * P.F, lane, r);
        int next = (r.slot + P.autoreset_stride) % P.bank_slots;
        int status = r.status, episodes = r.episodes + 1;
        reset_regs(P, next, r);
        r.status = status; r.episodes = episodes;
        acc = 0.0;
    } else if (done) {
        r.episodes += 1;
    }
    write_obs(P, obs_row, lane, r);
    if (lane < 3) P.acc[(size_t)env * 3 + lane] = acc;
    store_env(P, env, lane, r);
}

// XCD-aware wave -> environment mapping: workgroup b runs on XCD b % 8 (observed, speed only);
// each XCD walks one contiguous eighth of the environments so that neighbouring state rows
// (which share cache lines) meet in the same L2.
struct EnvWalker {
    int lo, hi, first, stride;
    __device__ __forceinline__ EnvWalker(int N, int waves_per_block) {
        const int nblk = gridDim.x;
        const int wave = rfl((int)(threadIdx.x >> 6));
        if (nblk % 8 == 0) {
            const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, nbx = nblk >> 3;
            lo = (int)(((long long)N * xcd) >> 3);
            hi = (int)(((long long)N * (xcd + 1)) >> 3);
            first = lo + bx * waves_per_block + wave;
            stride = nbx * waves_per_block;
        } else {
            lo = 0; hi = N;
            first = blockIdx.x * waves_per_block + wave;
            stride = nblk * waves_per_block;
        }
    }
};

// ------------------------------------------------------------------------------------------
// main step kernel: handles every environment whose projection is the box clip (always the case
// with project_action_in_env=False); the others are queued for the solver kernel.
// ------------------------------------------------------------------------------------------
template <bool PROJECT>
__global__ __launch_bounds__(256) void step_kernel(Params P, StepIO io) {
    __shared__ LdsNet net;
    stage_net(net, P);
    const int lane = threadIdx.x & 63;
    const int n = P.n, m = P.m, G = P.G;
    EnvWalker walk(P.N, 4);
    for (int env = walk.first; env < walk.hi; env += walk.stride) {
        EnvRegs r;
        load_env(P, env, lane, r);
        bool clamped;
        const double a = load_action(P, io, env, lane, clamped);
        if (r.t >= EVC_EPISODE_STEPS) {            // step() after termination without autoreset
            if (lane == 0) {
                P.scal[2 * env + 1].z = r.status | EVC_STATUS_STEP_AFTER_DONE;
                io.out.reward[env] = 0.0;
                io.out.terminated[env] = 1;
            }
            continue;
        }
        double y = a * Consts::ACTION_SCALE_FACTOR;     // env.py:366
        if (PROJECT) {
            // box part of the projection (exact when no network constraint binds)
            const double h = demand_cap_amps(r);
            y = fmin(y, h);
            // 1) conservative integer test: S_g <= sum ceil(8 y)/8, |w_c| <= sum_g |A_cg| S_g
            BitPlanes<9> planes;
            planes.build((int)ceil(y * 8.0));
            bool maybe = false;
            if (lane < m) {
                double bound = 0.0;
                for (int g = 0; g < G; g++)
                    bound += net.Aabs[g][lane] * (double)planes.sum(P.group_mask[g]);
                maybe = bound * 0.125 > net.mag[lane];
            }
            if (__ballot(maybe) != 0ull) {
                // 2) exact float64 class sums and row magnitudes
                double re = 0.0, im = 0.0;
                for (int g = 0; g < G; g++) {
                    const bool in_g = (P.group_mask[g] >> lane) & 1ull;
                    const double S = wave_sum_f64(in_g ? y : 0.0);
                    if (lane < m) { re += net.Mre[g][lane] * S; im += net.Mim[g][lane] * S; }
                }
                bool viol = false;
                if (lane < m)
                    viol = sqrt(re * re + im * im) > net.mag[lane] * (1.0 + Consts::PROJ_TOL);
                if (__ballot(viol) != 0ull) {
                    if (lane == 0) P.slow_list[atomicAdd(P.slow_count, 1)] = env;
                    continue;                      // the solver kernel steps this environment
                }
            }
        }
        finish_step(P, io, net, env, lane, y, y / Consts::ACTION_SCALE_FACTOR, clamped, r);
    }
}

// ------------------------------------------------------------------------------------------
// reset kernel (env.py:293-338): one wave per listed environment.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reset_kernel(Params P, const int* env_ids, const int* slots,
                                                    int count, float* obs) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= count) return;
    const int env = env_ids ? rfl(env_ids[w]) : w;
    const int slot = slots ? rfl(slots[w]) : (env % P.bank_slots);
    EnvRegs r;
    r.status = 0; r.episodes = 0;
    if (env < P.N) {
        const int4 s1 = P.scal[2 * env + 1];
        r.status = rfl(s1.z); r.episodes = rfl(s1.w);
    }
    reset_regs(P, slot, r);
    if (lane < 3) P.acc[(size_t)env * 3 + lane] = 0.0;
    if (obs) write_obs(P, obs + (size_t)env * P.F, lane, r);
    store_env(P, env, lane, r);
}

// metrics reduction: sums of the running accumulators + status census.
__global__ __launch_bounds__(256) void metrics_kernel(Params P, double* out) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, bad = 0.0, eps = 0.0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < P.N; e += gridDim.x * blockDim.x) {
        s0 += P.acc[(size_t)e * 3 + 0];
        s1 += P.acc[(size_t)e * 3 + 1];
        s2 += P.acc[(size_t)e * 3 + 2];
        const int4 s = P.scal[2 * e + 1];
        bad += (s.z != 0) ? 1.0 : 0.0;
        eps += (double)s.w;
    }
    s0 = wave_sum_f64(s0); s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2);
    bad = wave_sum_f64(bad); eps = wave_sum_f64(eps);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], s0); atomicAdd(&out[1], s1); atomicAdd(&out[2], s2);
        atomicAdd(&out[4], eps); atomicAdd(&out[5], bad);
    }
}

}  // namespace evc
