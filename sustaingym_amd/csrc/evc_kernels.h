// evc_kernels.h — the step / reset kernels (gfx950, wave-per-environment).
//
// Instruction budget: at the HBM roofline one environment-step may cost ~230 issue cycles per CU
// (65 536 envs, 151 MB, 8 TB/s, 256 CUs), so the streaming kernel is written for instruction
// count: no fp64 divides, class sums as two 16-bit fields per 32-bit DPP scan (6 fused
// v_add_u32_dpp per word), constraint rows screened in float32 with a safety margin and
// re-evaluated in float64 only when the screen is inconclusive, nothing on the (single per CU)
// scalar unit that can live on the four SIMDs.
#pragma once

#include "evc_device.h"

namespace evc {

// LDS image of the network tables needed by one workgroup (only the [G][m] corner).
struct LdsNet {
    double Mre[EVC_MAX_GROUPS][EVC_MAX_CONSTRAINTS];
    double Mim[EVC_MAX_GROUPS][EVC_MAX_CONSTRAINTS];
    double mag[EVC_MAX_CONSTRAINTS];
    float Mre32[EVC_MAX_GROUPS][EVC_MAX_CONSTRAINTS];
    float Mim32[EVC_MAX_GROUPS][EVC_MAX_CONSTRAINTS];
    float thr_y2[EVC_MAX_CONSTRAINTS];
    float thr_p2[EVC_MAX_CONSTRAINTS];
    float thr_yp2[EVC_MAX_CONSTRAINTS];
};

__device__ __forceinline__ void stage_net(LdsNet& s, const Params& P) {
    const int m = P.m;
    // classes are padded to an even count (two per packed word); padded entries are zero
    const int Gp = (P.G + 1) & ~1;
    for (int idx = threadIdx.x; idx < Gp * m; idx += blockDim.x) {
        int g = idx / m, c = idx - g * m;
        s.Mre[g][c] = P.tables->Mre[g][c];
        s.Mim[g][c] = P.tables->Mim[g][c];
        s.Mre32[g][c] = P.tables->Mre32[g][c];
        s.Mim32[g][c] = P.tables->Mim32[g][c];
    }
    for (int c = threadIdx.x; c < m; c += blockDim.x) {
        s.mag[c] = P.tables->mag[c];
        s.thr_y2[c] = P.tables->thr_y2[c];
        s.thr_p2[c] = P.tables->thr_p2[c];
        s.thr_yp2[c] = P.tables->thr_yp2[c];
    }
    __syncthreads();
}

// What the rarely taken projection branch of the streaming kernels needs of Params, held in LDS (stage_rare): read from
// the kernel-argument segment there, every field is a scalar load that misses the scalar cache — measured on the
// benchmark's day, 190 visits per step of ~5 us each, most of it such round trips, and the launch ends with the last visit.
struct LdsRare {
    double class_cap[EVC_MAX_GROUPS];
    double snap_tol;
    unsigned simple_rows, cap_classes;
    int monotone_rows, G, tie_log2;
};
__device__ __forceinline__ void stage_rare(LdsRare& r, const Params& P) {          // before a workgroup barrier
    // one lane, uniform indices: the fields arrive through the scalar loads the kernel's other arguments take anyway (a
    // lane-indexed copy is a VECTOR load from the kernel-argument segment: +0.4 us of prologue, measured)
    if (threadIdx.x == 0) {
#pragma unroll
        for (int g = 0; g < EVC_MAX_GROUPS; g++) r.class_cap[g] = P.class_cap[g];
        r.snap_tol = P.snap_tol; r.simple_rows = P.simple_rows; r.cap_classes = P.cap_classes;
        r.monotone_rows = P.monotone_rows; r.G = P.G; r.tie_log2 = P.tie_log2;
    }
}

// Per-lane constants of the network (computed once per wave).
struct LaneNet {
    int gid;          // station class of this lane, -1 outside the network
    int word;         // packed word holding this class' sum (gid >> 1)
    int shift;        // 0 or 16
    bool is_cc;
    bool in_net;
};

__device__ __forceinline__ LaneNet lane_net(const Params& P, int lane) {
    LaneNet ln;
    ln.in_net = lane < P.n;
    ln.gid = -1;
    for (int g = 0; g < P.G; g++)
        if ((P.group_mask[g] >> lane) & 1ull) ln.gid = g;
    ln.word = ln.gid >> 1;
    ln.shift = (ln.gid & 1) << 4;
    ln.is_cc = (P.cc_mask >> lane) & 1ull;
    return ln;
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

// Raw loads of one environment's rows.
struct EnvLoads {
    int4 s0, s1;      // env scalars (uniform address, broadcast)
    double rem;       // remaining-demand row element
    int de;           // packed departure / est_departure
    float a;          // action element
    double acc;       // breakdown accumulator (lanes < 3)
};

// wave_in_block: which of the workgroup's (up to 4) wavefronts calls; -1 = derive it from the thread id (the slow path, always
// wave 0, says so: it is also reached through a call whose callee should not need work-item ids from its caller)
__device__ __forceinline__ EnvLoads issue_loads(const Params& P, const StepIO& io, int env, int lane, int wave_in_block = -1) {
    EnvLoads L;
    const unsigned n = (unsigned)P.n, ul = (unsigned)lane;
    L.s0 = P.scal[2 * env];
    L.s1 = P.scal[2 * env + 1];
    if (P.compact) {
        // entries -> stations: every entry announces its lane at its station's cell, then each
        // station lane pulls the entry's words with a lane gather (ds_bpermute)
        __shared__ int entry_of[4][kWave];
        int* cell = entry_of[wave_in_block >= 0 ? wave_in_block : (int)(threadIdx.x >> 6)];
        // The entry rows are requested beside the scalars (bound: the whole row; entries beyond the count A are masked by
        // `lane < A` below) instead of behind them through a descriptor of A entries: one dependent round trip fewer per
        // queued environment — the slow path is memory latency, profiles/r3_solver_stats.txt (JPL GMM day 47.1 -> 46.8 us
        // per step pipelined, Caltech 33.3 -> 33.0; profiles/r3_fill_ab.txt).
        const double rem_e = buf_ld_f64(row_rsrc(P.rem + (size_t)env * n, n * 8u), ul * 8u);
        const unsigned w_e = buf_ld_u32(row_rsrc(P.depest + (size_t)env * n, n * 4u), ul * 4u);
        const unsigned A = (unsigned)(rfl(L.s1.z) >> kCountShift) & 0x7fu;
        // (LDS executes a wave's ds instructions in issue order: compiler barriers suffice)
        cell[lane] = 0;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (ul < A) cell[entry_station(w_e)] = lane + 1;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int src = cell[lane] - 1;
        const int from = src < 0 ? 0 : src;
        const unsigned w_d = (unsigned)__shfl((int)w_e, from);
        const double rem_d = __shfl(rem_e, from);
        L.rem = src < 0 ? 0.0 : rem_d;
        L.de = src < 0 ? (kEmptyDep & 0xffff) : (int)((w_d & 0x3ffu) | (w_d & 0xffff0000u));
    } else {
        L.rem = buf_ld_f64(row_rsrc(P.rem + (size_t)env * n, n * 8u), ul * 8u);
        L.de = (int)buf_ld_u32(row_rsrc(P.depest + (size_t)env * n, n * 4u), ul * 4u);
    }
    L.a = (io.actions && io.action_kind == EVC_ACTION_F32) ? buf_ld_f32(row_rsrc((const float*)io.actions + (size_t)env * n, n * 4u), ul * 4u) : 0.0f;
    L.acc = buf_ld_f64(row_rsrc(P.acc + (size_t)env * 3, 24u), ul * 8u);
    if (ul >= n) L.de = kEmptyDep & 0xffff;          // lanes outside the network: empty EVSE
    return L;
}

__device__ __forceinline__ void unpack_env(const EnvLoads& L, EnvRegs& r) {
    r.t = rfl(L.s0.x); r.cursor = rfl(L.s0.y); r.slot = rfl(L.s0.z); r.moer_day = rfl(L.s0.w);
    r.n_sessions = rfl(L.s1.x); r.next_arrival = rfl(L.s1.y); r.status = rfl(L.s1.z) & kStatusMask; r.episodes = rfl(L.s1.w);
    r.rem = L.rem;
    r.dep = (int)(short)(L.de & 0xffff);
    r.est = L.de >> 16;
}

// Normalised action clamped to [0,1] (the reference raises on out-of-range actions, SURVEY §8a
// a2), as float64.
__device__ __forceinline__ double unpack_action(const StepIO& io, const EnvLoads& L, bool& clamped) {
    float a = L.a;
    if (io.action_kind == EVC_ACTION_GREEDY) {            // baselines.py:32-35 on the observation
        const int dep = (int)(short)(L.de & 0xffff);
        a = (dep != kEmptyDep && L.rem > Consts::FULLY_CHARGED_EPS) ? 1.0f : 0.0f;
    }
    clamped = !(a >= 0.0f && a <= 1.0f);                 // also true for NaN
    a = fminf(fmaxf(a, 0.0f), 1.0f);                      // NaN -> 0
    return (double)a;
}

__device__ __forceinline__ void load_env(const Params& P, int env, int lane, EnvRegs& r) {
    StepIO none{};
    none.actions = nullptr;
    unpack_env(issue_loads(P, none, env, lane), r);
}

__device__ __forceinline__ void store_env(const Params& P, int env, int lane, const EnvRegs& r) {
    const unsigned n = (unsigned)P.n, ul = (unsigned)lane;
    int status = r.status & kStatusMask;
    if (P.compact) {
        const bool occ = ul < n && r.dep != kEmptyDep;
        const unsigned long long mask = __ballot(occ);
        const unsigned pos = occ ? (unsigned)__popcll(mask & ((1ull << lane) - 1ull)) : 0x1fffffffu;
        buf_st_f64(row_rsrc(P.rem + (size_t)env * n, n * 8u), pos * 8u, r.rem);
        buf_st_u32(row_rsrc(P.depest + (size_t)env * n, n * 4u), pos * 4u, pack_entry(r.dep, lane, r.est));
        status |= __popcll(mask) << kCountShift;
    } else {
        buf_st_f64(row_rsrc(P.rem + (size_t)env * n, n * 8u), ul * 8u, r.rem);
        buf_st_u32(row_rsrc(P.depest + (size_t)env * n, n * 4u), ul * 4u,
                   (unsigned)((r.dep & 0xffff) | (r.est << 16)));
    }
    // the two int4 of scalars: only lane 0 is inside the 16-byte windows
    const rsrc_t s = row_rsrc(P.scal + 2 * (size_t)env, 32u);
    buf_st_i4(s, ul * 32u, make_int4(r.t, r.cursor, r.slot, r.moer_day));
    buf_st_i4(s, ul * 32u + 16u, make_int4(r.n_sessions, r.next_arrival, status, r.episodes));
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

// MOER part of the observation row for lane j < k+2: [forecast_1..k | prev_moer | t/288]
__device__ __forceinline__ float moer_obs_value(const Params& P, int lane, int moer_day, int t) {
    const float* mrow = P.moer_obs + ((size_t)moer_day * EVC_MOER_ROWS + t) * EVC_MOER_COLS;
    const unsigned idx = (lane < P.k) ? (unsigned)lane + 1u : 0u;             // env.py:390-391
    float v = buf_ld_f32(row_rsrc(mrow, EVC_MOER_COLS * 4u), idx * 4u);
    const float ts = P.tables->timestep[t];             // (float)(t / 288.0), env.py:392, host table
    return (lane == P.k + 1) ? ts : v;
}

// env.py:381-394: observation row [demands | est_departures | forecasted_moer | prev_moer | t/288]
__device__ __forceinline__ void write_obs(const Params& P, float* row, int lane, const EnvRegs& r,
                                          float moer_value) {
    const unsigned n = (unsigned)P.n, ul = (unsigned)lane;
    const bool active = (r.dep != kEmptyDep) && (r.rem > Consts::FULLY_CHARGED_EPS);
    buf_st_f32(row_rsrc(row, n * 4u), ul * 4u, active ? (float)r.rem : 0.0f);
    buf_st_f32(row_rsrc(row + n, n * 4u), ul * 4u, active ? (float)(r.est - r.t) : 0.0f);
    buf_st_f32(row_rsrc(row + 2 * n, (unsigned)(P.k + 2) * 4u), ul * 4u, moer_value);
}

// Upper bound of the projected action (amps): min(32, demand_f32 / A_PERS_TO_KWH)
// (env.py:188-189 with demands = previous float32 observation, env.py:218).
__device__ __forceinline__ double demand_cap_amps(const EnvRegs& r) {
    const bool active = (r.dep != kEmptyDep) && (r.rem > Consts::FULLY_CHARGED_EPS);
    const double demand = (double)(float)r.rem;
    const double cap = fmin(demand / Consts::A_PERS_TO_KWH, Consts::ACTION_SCALE_FACTOR);
    return active ? cap : 0.0;
}

// Two 16-bit class sums per 32-bit word: every lane contributes `q` (< 2^16 / 64) to the field
// of its own class; WORDS DPP scans give all class sums of the environment.
template <int WORDS>
__device__ __forceinline__ void class_sums_u16(const LaneNet& ln, unsigned q, unsigned (&tot)[WORDS]) {
    const unsigned qs = q << ln.shift;
#pragma unroll
    for (int w = 0; w < WORDS; w++) tot[w] = wave_total_u32(ln.word == w ? qs : 0u);
}

// |M_c S|^2 in float32 for row c = lane (screening only).
template <int WORDS>
__device__ __forceinline__ float row_mag2_f32(const LdsNet& net, int c, const unsigned (&tot)[WORDS]) {
    float re = 0.0f, im = 0.0f;
#pragma unroll
    for (int w = 0; w < WORDS; w++) {
        const float s0 = (float)(tot[w] & 0xffffu), s1 = (float)(tot[w] >> 16);
        re = fmaf(net.Mre32[2 * w][c], s0, re);
        im = fmaf(net.Mim32[2 * w][c], s0, im);
        re = fmaf(net.Mre32[2 * w + 1][c], s1, re);
        im = fmaf(net.Mim32[2 * w + 1][c], s1, im);
    }
    return re * re + im * im;
}

// |M_c S| in float64 for row c = lane from exact integer class sums (scaled by `scale`).
template <int WORDS>
__device__ __forceinline__ double row_mag_f64(const LdsNet& net, int c, const unsigned (&tot)[WORDS],
                                              double scale) {
    double re = 0.0, im = 0.0;
#pragma unroll
    for (int w = 0; w < WORDS; w++) {
        const double s0 = (double)(tot[w] & 0xffffu) * scale, s1 = (double)(tot[w] >> 16) * scale;
        re += net.Mre[2 * w][c] * s0 + net.Mre[2 * w + 1][c] * s1;
        im += net.Mim[2 * w][c] * s0 + net.Mim[2 * w + 1][c] * s1;
    }
    return sqrt(re * re + im * im);
}

// Everything of EVChargingEnv.step after the projection: rounding to legal pilots, one
// acnsim.Simulator.step pass, observation, reward, bookkeeping, autoreset.
//   y : this lane's (projected) action in amps, before rounding
// Ordering is for instruction-level overlap inside the in-order wave: the reductions (DPP
// ladders) are issued first, everything that does not depend on them (event pass, observation
// and state stores) follows, and the only data-dependent branch (exact constraint excess, rare)
// is taken last.
template <int WORDS>
__device__ __forceinline__ void finish_step(const Params& P, const StepIO& io, const LdsNet& net,
                                            const LaneNet& ln, int env, int lane, double y,
                                            bool clamped, double acc, bool pilots_screened,
                                            EnvRegs& r) {
    const unsigned n = (unsigned)P.n;
    const int m = P.m;
    const int t1 = r.t + 1;                                         // env.py:279
    float moer_value = moer_obs_value(P, lane, r.moer_day, t1);      // issue the loads early
    const double moer_now = P.moer_hist[(size_t)r.moer_day * EVC_MOER_ROWS + t1];

    // ---- env.py:366-378 pilots ----
    const double pilot = ln.in_net ? legal_pilot(y, ln.is_cc) : 0.0;

    // ---- acnsim update_pilots: charge the plugged EVs for iteration t1-1 ----
    const bool occupied = r.dep != kEmptyDep;
    const double amps = charge_ev(occupied ? pilot : 0.0, r.rem, P.battery_stepwise != 0);
    const double total_rate = wave_sum_f64(amps);                    // env.py:445

    // ---- env.py:449-452 screen of the PILOT schedule (exact integer class sums, float32 rows) ----
    unsigned ptot[WORDS];
    unsigned long long maybe_rows = 0ull;
    if (!pilots_screened) {                 // (wave-uniform) the y screen did not already clear them
        class_sums_u16<WORDS>(ln, (unsigned)(int)pilot, ptot);
        bool maybe = false;
        if (lane < m) maybe = !(row_mag2_f32<WORDS>(net, lane, ptot) < net.thr_p2[lane]);
        maybe_rows = __ballot(maybe);
    }

    // ---- optional per-station debug outputs ----
    if (io.out.pilots) buf_st_f64(row_rsrc(io.out.pilots + (size_t)env * n, n * 8u), lane * 8u, pilot);
    if (io.out.rates) buf_st_f64(row_rsrc(io.out.rates + (size_t)env * n, n * 8u), lane * 8u, amps);
    if (io.out.projected)
        buf_st_f64(row_rsrc(io.out.projected + (size_t)env * n, n * 8u), lane * 8u,
                   y / Consts::ACTION_SCALE_FACTOR);

    // ---- acnsim event pass at iteration t1: unplug (precedence 0) before plug-in (10) ----
    if (occupied && r.dep <= t1) { r.dep = kEmptyDep; r.est = 0; r.rem = 0.0; }
    // The sessions from the cursor on, one per lane, in ONE round trip (a morning burst plugs in several EVs in the same
    // period; fetched one by one each costs two dependent loads); refilled after 63 arrivals.
    while (r.next_arrival <= t1 && r.cursor < r.n_sessions) {
        const int base = r.cursor;
        const size_t s0 = (size_t)r.slot * P.max_sessions + base;
        const bool inb = base + lane < r.n_sessions;
        unsigned long long sw = ~0ull;                // arrival -1 -> never reached
        double rql = 0.0;
        if (inb) {
            sw = *reinterpret_cast<const unsigned long long*>(P.sessions + s0 + lane);
            rql = P.requested[s0 + lane];
        }
        while (r.next_arrival <= t1 && r.cursor < r.n_sessions && r.cursor - base < 63) {
            const int k = r.cursor - base;
            const unsigned long long w = (unsigned long long)(unsigned)__shfl((int)(unsigned)sw, k) |
                                         ((unsigned long long)(unsigned)__shfl((int)(unsigned)(sw >> 32), k) << 32);
            const double rq = __shfl(rql, k);
            const int s_departure = (int)(short)(w >> 16), s_est = (int)(short)(w >> 32);
            const int st = rfl((int)(short)(w >> 48));
            const bool mine = lane == st;
            const bool busy = mine && (r.dep != kEmptyDep);
            if (__ballot(busy) != 0ull) {
                r.status |= EVC_STATUS_OCCUPIED;          // acnportal: StationOccupiedError
            } else if (mine) {
                r.dep = s_departure; r.est = s_est; r.rem = rq;
            }
            r.cursor += 1;
            const int nxt = rfl((int)(short)(unsigned short)__shfl((int)(unsigned)sw, k + 1));   // k + 1 <= 63
            r.next_arrival = (r.cursor < r.n_sessions) ? nxt : kNoArrival;
        }
    }
    r.t = t1;
    if (__ballot(clamped && ln.in_net) != 0ull) r.status |= EVC_STATUS_ACTION_CLAMPED;
    const bool done = t1 >= EVC_EPISODE_STEPS;     // event queue empty after the pass at 288

    // ---- observation + state write-back (independent of the reductions) ----
    float* obs_row = io.out.obs + (size_t)env * P.F;
    EnvRegs w = r;                                  // state to store (reset state on autoreset)
    if (done) {
        r.episodes += 1;
        w.episodes = r.episodes;
        if (P.autoreset) {
            if (io.out.final_obs) write_obs(P, io.out.final_obs + (size_t)env * P.F, lane, r, moer_value);
            const int next = (r.slot + P.autoreset_stride) % P.bank_slots;
            reset_regs(P, next, w);
            w.status = r.status; w.episodes = r.episodes;
            moer_value = moer_obs_value(P, lane, w.moer_day, 0);
        }
    }
    write_obs(P, obs_row, lane, w, moer_value);
    store_env(P, env, lane, w);

    // ---- env.py:431-464 reward (needs the reductions) ----
    double excess = 0.0;
    if (maybe_rows != 0ull) {                                        // rare: evaluate exactly
        double ex = 0.0;
        if (lane < m) ex = fmax(row_mag_f64<WORDS>(net, lane, ptot, 1.0) - net.mag[lane], 0.0);
        excess = wave_sum_f64(ex);
    }
    const double profit = Consts::PROFIT_FACTOR * total_rate;
    const double carbon = Consts::CARBON_COST_FACTOR * total_rate * moer_now;
    const double excess_charge = excess * Consts::VIOLATION_FACTOR;
    const double reward = profit - carbon - excess_charge;
    acc += (lane == 0) ? profit : (lane == 1 ? carbon : excess_charge);
    buf_st_f64(row_rsrc(io.out.reward + env, 8u), lane * 8u, reward);              // lane 0 only
    if (io.out.returns && lane == 0) io.out.returns[env] += reward;
    buf_st_u8(row_rsrc(io.out.terminated + env, 1u), lane, done ? 1 : 0);            // lane 0 only
    if (io.out.breakdown) buf_st_f64(row_rsrc(io.out.breakdown + (size_t)env * 3, 24u), lane * 8u, acc);
    buf_st_f64(row_rsrc(P.acc + (size_t)env * 3, 24u), lane * 8u, (done && P.autoreset) ? 0.0 : acc);
    r = w;
}

// Closed-form projection for a violated "simple" row (a cap on one station class, e.g. a pod
// breaker): find nu >= 0 with  sum_{i in class} clip(b_i - nu, 0, h_i) = cap  and return the
// clipped values.  Safeguarded Newton on the piecewise-linear sum (bracketed; on a flat piece it
// jumps to the next breakpoint).  Wave-uniform control flow; lanes outside the class pass through.
__device__ __forceinline__ double waterfill_class(bool in_g, double b, double h, double cap, double y) {
    double nu = 0.0, lo = 0.0, hi = 64.0;
    for (int it = 0; it < 80; it++) {
        const double v = b - nu;
        const double yv = in_g ? fmin(fmax(v, 0.0), h) : 0.0;
        const bool is_free = in_g && v > 0.0 && v <= h && h > 0.0;
        const double f = wave_sum_f64(yv) - cap;
        if (fabs(f) <= 1e-13 * cap) break;
        if (f > 0.0) lo = nu; else hi = nu;
        const int kfree = __popcll(__ballot(is_free));
        double nxt;
        if (kfree > 0) {
            nxt = nu + f / (double)kfree;
        } else if (f > 0.0) {       // flat piece above the cap: move to the next breakpoint
            nxt = wave_min_f64((in_g && v > h) ? b - h : 1e300);
        } else {
            nxt = 0.5 * (lo + hi);
        }
        if (!(nxt > lo && nxt < hi)) nxt = 0.5 * (lo + hi);
        nu = nxt;
    }
    return in_g ? fmin(fmax(b - nu, 0.0), h) : y;
}

// exact float64 feasibility of the schedule y: returns the ballot of violated rows and the
// bitmask of classes whose simple-row cap is exceeded
template <typename Net>
__device__ __forceinline__ unsigned long long exact_rows(const Params& P, const Net& net, const LaneNet& ln,
                                                         int lane, double y, unsigned& cap_viol) {
    double re = 0.0, im = 0.0;
    cap_viol = 0u;
    for (int g = 0; g < P.G; g++) {
        const double S = wave_sum_f64(ln.gid == g ? y : 0.0);
        if (lane < P.m) { re += net.Mre[g][lane] * S; im += net.Mim[g][lane] * S; }
        if (S > P.class_cap[g] * (1.0 + Consts::PROJ_TOL)) cap_viol |= 1u << g;
    }
    bool viol = false;
    if (lane < P.m) viol = sqrt(re * re + im * im) > net.mag[lane] * (1.0 + Consts::PROJ_TOL);
    return __ballot(viol);
}

// Queue control block of one step (double-buffered by step parity): ctl[0] = number of queued environments.
__device__ __forceinline__ int queue_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void queue_push(const Params& P, int env) {
    const int idx = __hip_atomic_fetch_add(P.slow_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // agent-coherent store: the draining workgroup may sit on another XCD (own L2) and read it in the SAME launch
    __hip_atomic_store(P.slow_list + idx, env, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// what every drainer does first: clear the other parity's control block for the next step, report the queue
// length to the host (page-locked ring the engine reads without synchronising, evc_engine.hip: drain mode)
__device__ __forceinline__ void queue_begin_drain(const Params& P, int count) {
    P.slow_count_next[0] = 0;
    if (P.host_qlen) __hip_atomic_store(P.host_qlen, count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// XCD-aware wave -> environment mapping: workgroup b runs on XCD b % 8 (observed, speed only);
// each XCD walks one contiguous eighth of the environments so that neighbouring state rows
// (which share cache lines) meet in the same L2.
struct EnvWalker {
    int hi, first, stride;
    __device__ __forceinline__ EnvWalker(int N, int waves_per_block) {
        const int nblk = gridDim.x;
        const int wave = rfl((int)(threadIdx.x >> 6));
#ifndef EVC_WALK_INTERLEAVED
#define EVC_WALK_INTERLEAVED 0
#endif
        if (!EVC_WALK_INTERLEAVED && nblk % 8 == 0) {
            const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, nbx = nblk >> 3;
            const int lo = (int)(((long long)N * xcd) >> 3);
            hi = (int)(((long long)N * (xcd + 1)) >> 3);
            first = lo + bx * waves_per_block + wave;
            stride = nbx * waves_per_block;
        } else {
            hi = N;
            first = blockIdx.x * waves_per_block + wave;
            stride = nblk * waves_per_block;
        }
    }
};

// ------------------------------------------------------------------------------------------
// main step kernel: handles every environment whose projection is the box clip (always the case
// with project_action_in_env=False); the others are queued for the solver kernel.
// ------------------------------------------------------------------------------------------
template <bool PROJECT, int WORDS>
__global__ __launch_bounds__(256) void step_kernel(Params P, StepIO io) {
    __shared__ LdsNet net;
    stage_net(net, P);
    const int lane = threadIdx.x & 63;
    const int m = P.m;
    const LaneNet ln = lane_net(P, lane);
    EnvWalker walk(P.N, 4);
    if (walk.first >= walk.hi) return;
    for (int env = walk.first; env < walk.hi; env += walk.stride) {
        const EnvLoads cur = issue_loads(P, io, env, lane);
        EnvRegs r;
        unpack_env(cur, r);
        bool clamped;
        const double a = unpack_action(io, cur, clamped);
        if (r.t >= EVC_EPISODE_STEPS) {            // step() after termination without autoreset
            if (lane == 0) {
                P.scal[2 * env + 1].z = cur.s1.z | EVC_STATUS_STEP_AFTER_DONE;   // keeps the entry count
                io.out.reward[env] = 0.0;
                io.out.terminated[env] = 1;
            }
            continue;
        }
        const double b = a * Consts::ACTION_SCALE_FACTOR;     // env.py:366
        double y = b;
        bool pilots_screened = false;
        if (PROJECT) {
            // box part of the projection (exact when no network constraint binds)
            const double h = demand_cap_amps(r);
            y = fmin(b, h);
            // 1) screen: class-sum upper bounds from ceil(8 y) (1/8 A units), float32 rows against
            //    thresholds that absorb the quantisation slack and the float32 error
            unsigned ytot[WORDS];
            class_sums_u16<WORDS>(ln, (unsigned)(int)ceil(y * 8.0), ytot);
            bool maybe = false, maybe_p = false;
            if (lane < m) {
                const float mag2 = row_mag2_f32<WORDS>(net, lane, ytot);
                maybe = !(mag2 < net.thr_y2[lane]);
                maybe_p = !(mag2 < net.thr_yp2[lane]);
            }
            pilots_screened = __ballot(maybe_p) == 0ull;   // rounded pilots cannot violate either
            if (__ballot(maybe) != 0ull) {
                // 2) exact float64 class sums and row magnitudes (rare)
                unsigned cap_viol;
                unsigned long long vrows = exact_rows(P, net, ln, lane, y, cap_viol);
                if (vrows != 0ull) {
                    bool solved = false;
                    if (cap_viol != 0u) {
                        // 3) class caps (pod breakers) are violated: closed-form water-filling; exact if
                        //    the result satisfies every row (relaxation argument), also when multi-class
                        //    rows were violated before the caps were applied
                        double yw = y;
                        for (int g = 0; g < P.G; g++)
                            if ((cap_viol >> g) & 1u)
                                yw = waterfill_class(ln.gid == g, b, h, P.class_cap[g], yw);
                        unsigned cv2;
                        if (exact_rows(P, net, ln, lane, yw, cv2) == 0ull) {
                            // tie snap of solver-moved values
                            if (yw != y) yw = tie_snap_counted(yw, h, ln.is_cc, P.tie_counters, P.tie_log2);
                            y = yw;
                            solved = true;
                        }
                    }
                    if (!solved) {
                        if (lane == 0) queue_push(P, env);
                        continue;                  // the solver kernel steps this environment
                    }
                }
            }
        }
        finish_step<WORDS>(P, io, net, ln, env, lane, y, clamped, cur.acc, pilots_screened, r);
    }
}

constexpr unsigned kPolicyTag = 0x504f4c43u;      // random policy: last counter word (see random_actions_kernel)

#ifndef EVC_TEMPLATES_ONLY      /* the non-template kernels belong to ONE translation unit (evc_engine.hip) */
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
    const int4 s1 = P.scal[2 * env + 1];
    r.status = rfl(s1.z) & kStatusMask; r.episodes = rfl(s1.w);
    reset_regs(P, slot, r);
    buf_st_f64(row_rsrc(P.acc + (size_t)env * 3, 24u), lane * 8u, 0.0);
    if (obs) write_obs(P, obs + (size_t)env * P.F, lane, r, moer_obs_value(P, lane, r.moer_day, 0));
    store_env(P, env, lane, r);
}

// DiscreteActionWrapper.action (wrappers.py:43-45): int64 {0..bins-1} -> float32 a/(bins-1)
// (float32 division, as numpy does), written to the engine's float32 action staging buffer.
__global__ __launch_bounds__(256) void discretize_kernel(const long long* in, float* out, size_t count,
                                                         int bins) {
    const float denom = (float)(bins - 1);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (float)in[i] / denom;
}

// Device-resident RandomAlgorithm policy (algorithms/evcharging/baselines.py:38-51): uniform actions for
// every environment at its CURRENT period, written to the [N][n] float32 action buffer the step kernel
// reads.  Counter-based (Philox4x32-10, key = 64-bit policy seed), so the action of (environment, episode,
// period, station) does not depend on launch geometry, sharding or call history:
//   counter = (t | block << 16, episodes_done, env_id_base + env, 0x504f4c43), block = station / 4,
//   word j -> station 4 block + j;  continuous: a = (w >> 8) 2^-24 in [0,1);  discrete (bins >= 2):
//   level = (w bins) >> 32, a = float(level) / float(bins - 1)  (wrappers.py:43-45).
// (include/evcharge.h: evc_set_policy_seed states the rule; the tests hold an independent scalar C statement of it.)
__global__ __launch_bounds__(256) void random_actions_kernel(const int4* __restrict__ scal, float* __restrict__ out,
                                                             int N, int n, int bins, unsigned long long seed,
                                                             unsigned env_id_base) {
    const int blocks_per_env = (n + 3) >> 2;
    const size_t total = (size_t)N * blocks_per_env;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int env = (int)(i / blocks_per_env), block = (int)(i % blocks_per_env);
        const int t = scal[(size_t)env * 2].x, episodes = scal[(size_t)env * 2 + 1].w;
        const Philox ph((unsigned)t | ((unsigned)block << 16), (unsigned)episodes, env_id_base + (unsigned)env, kPolicyTag,
                        (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int s = block * 4 + j;
            if (s >= n) break;
            const unsigned w = ph.w[j];
            float a;
            if (bins >= 2) a = (float)(unsigned)(((unsigned long long)w * (unsigned)bins) >> 32) / (float)(bins - 1);
            else a = (float)(w >> 8) * (1.0f / 16777216.0f);
            out[(size_t)env * n + s] = a;
        }
    }
}

// Per-agent observation gather (multiagent_env.py:102-148), HBM-bound: one workgroup per
// environment streams n rows of F floats (31.5 kB for Caltech) from two F-float source rows.
__global__ __launch_bounds__(256) void gather_agent_obs_kernel(const float* __restrict__ obs,
                                                               const float* __restrict__ delayed,
                                                               float* __restrict__ out, int N, int n, int F) {
    for (int env = blockIdx.x; env < N; env += gridDim.x) {
        const float* cur = obs + (size_t)env * F;
        const float* old = delayed ? delayed + (size_t)env * F : cur;
        float* dst = out + (size_t)env * n * F;
        const int total = n * F;
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            const int a = i / F, f = i - a * F;
            const bool others = f < 2 * n && f != a && f != n + a;   // another agent's demand / est_departure
            dst[i] = others ? old[f] : cur[f];
        }
    }
}

// The same for even F (the usual case: F = 2n + k + 2 with k = 36), HBM-write-bound as it should be: a thread owns ONE
// pair of columns (f, f+1) — its two current and two delayed values sit in registers for the whole environment — and
// walks the agents: column f belongs to "another agent" for every agent except a == f (demands) / a == f - n
// (est_departures), so a row element is one compare + select, and every store is a coalesced 8-byte store (rows of
// F floats are 8-byte aligned).  No per-element division, no per-element source load: 66 -> see DESIGN.md §6.
__global__ __launch_bounds__(256) void gather_agent_obs_pairs_kernel(const float2* __restrict__ obs,
                                                                     const float2* __restrict__ delayed,
                                                                     float2* __restrict__ out, int N, int n, int F) {
    const int P2 = F >> 1;                          // column pairs per row
    const int groups = (int)blockDim.x / P2;        // agents handled side by side
    const int p = (int)threadIdx.x % P2, g = (int)threadIdx.x / P2;
    if (g >= groups) return;
    const int f0 = 2 * p, f1 = f0 + 1;
    // agent for which column f is its OWN entry (-1: never, e.g. the MOER / timestep tail)
    const int own0 = f0 < n ? f0 : (f0 < 2 * n ? f0 - n : -1), own1 = f1 < n ? f1 : (f1 < 2 * n ? f1 - n : -1);
    for (int env = blockIdx.x; env < N; env += gridDim.x) {
        const float2 cur = obs[(size_t)env * P2 + p];
        float2 old = delayed ? delayed[(size_t)env * P2 + p] : cur;
        if (own0 < 0) old.x = cur.x;                // columns beyond 2n are never delayed
        if (own1 < 0) old.y = cur.y;
        float2* dst = out + (size_t)env * n * P2 + p;
        for (int a = g; a < n; a += groups) {
            float2 v;
            v.x = a == own0 ? cur.x : old.x;
            v.y = a == own1 ? cur.y : old.y;
            dst[(size_t)a * P2] = v;
        }
    }
}

// metrics reduction: sums of the running accumulators + status census.
__global__ __launch_bounds__(256) void metrics_kernel(Params P, double* out) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, bad = 0.0, eps = 0.0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < P.N; e += gridDim.x * blockDim.x) {
        s0 += P.acc[(size_t)e * 3 + 0];
        s1 += P.acc[(size_t)e * 3 + 1];
        s2 += P.acc[(size_t)e * 3 + 2];
        const int4 s = P.scal[2 * e + 1];
        bad += ((s.z & kStatusMask) != 0) ? 1.0 : 0.0;
        eps += (double)s.w;
    }
    s0 = wave_sum_f64(s0); s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2);
    bad = wave_sum_f64(bad); eps = wave_sum_f64(eps);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], s0); atomicAdd(&out[1], s1); atomicAdd(&out[2], s2);
        atomicAdd(&out[4], eps); atomicAdd(&out[5], bad);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double moved = 0.0, near = 0.0;
        for (int i = 0; i < kTieSlots; i++) { moved += (double)P.tie_counters[2 * i]; near += (double)P.tie_counters[2 * i + 1]; }
        out[6] = moved;
        out[7] = near;
    }
}

#endif  // EVC_TEMPLATES_ONLY

}  // namespace evc
