// evc_quad.h — the streaming step kernel: FOUR environments per wavefront.
//
// Layout: wavefront = 4 DPP rows of 16 lanes; row r simulates environment 4*quad + r, lane q of a
// row owns stations q, q+16, q+32, q+48 ("slots" j = 0..3, n <= 64).  What is uniform per
// environment (t, cursor, bank slot, MOER day, status, reward terms) is replicated across the 16
// lanes of its row in VGPRs; cross-station reductions are 4-step DPP `row_ror` all-reduces that
// never leave the row.  Compared with one environment per wavefront this amortises all per-
// environment work (reductions, constraint rows, scalar bookkeeping, address generation) over four
// environments and removes almost all scalar-unit work; per-station work is unchanged.  Rows whose
// projection screen is inconclusive are appended to the slow queue and finished by the
// wave-per-environment kernel (evc_solver.h); they write nothing here.
//
// Requires m <= 16 (constraint row c is evaluated by lane c of every row).
//
// Template parameter DBG selects the generic instantiation (per-station debug outputs, the
// device-resident greedy policy, the returns accumulator — all checked at run time); DBG = false
// is the lean streaming kernel used by plain evc_step calls.
#pragma once

#include "evc_kernels.h"

namespace evc {

constexpr int kSlots = 4;

template <int CTRL>
__device__ __forceinline__ unsigned dpp_ror_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned row_allreduce_u32(unsigned v) {
    v += dpp_ror_u32<0x121>(v);   // row_ror:1
    v += dpp_ror_u32<0x122>(v);   // row_ror:2
    v += dpp_ror_u32<0x124>(v);   // row_ror:4
    v += dpp_ror_u32<0x128>(v);   // row_ror:8
    return v;
}
// The packed class-sum words of the streaming kernels, all of them through the four rotation steps together (round 6).
// Written as v_add_u32_dpp by hand: the compiler fuses only some steps of row_allreduce_u32 into the add (the first one
// disappears into the v_mad_u32_u24 that forms the word, the last one is kept as v_mov 0 + v_mov_dpp + v_add: 36 vector
// instructions for three words instead of 12).  Words go in groups of up to three, interleaved: a DPP source written by a
// vector instruction needs two wait states, which the other words of the group provide (s_nop where they do not, and in
// front of the first step, whose inputs the compiler's own instructions have just written).
#define EVC_DPP_ADD_(i, ctrl) "v_add_u32_dpp %" #i ", %" #i ", %" #i " " ctrl " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void row_allreduce_u32x3(unsigned& a, unsigned& b, unsigned& c) {
    asm("s_nop 1\n\t" EVC_DPP_ADD_(0, "row_ror:1") EVC_DPP_ADD_(1, "row_ror:1") EVC_DPP_ADD_(2, "row_ror:1")
        EVC_DPP_ADD_(0, "row_ror:2") EVC_DPP_ADD_(1, "row_ror:2") EVC_DPP_ADD_(2, "row_ror:2")
        EVC_DPP_ADD_(0, "row_ror:4") EVC_DPP_ADD_(1, "row_ror:4") EVC_DPP_ADD_(2, "row_ror:4")
        EVC_DPP_ADD_(0, "row_ror:8") EVC_DPP_ADD_(1, "row_ror:8") EVC_DPP_ADD_(2, "row_ror:8")
        : "+v"(a), "+v"(b), "+v"(c));
}
__device__ __forceinline__ void row_allreduce_u32x2(unsigned& a, unsigned& b) {
    asm("s_nop 1\n\t" EVC_DPP_ADD_(0, "row_ror:1") EVC_DPP_ADD_(1, "row_ror:1") "s_nop 0\n\t"
        EVC_DPP_ADD_(0, "row_ror:2") EVC_DPP_ADD_(1, "row_ror:2") "s_nop 0\n\t"
        EVC_DPP_ADD_(0, "row_ror:4") EVC_DPP_ADD_(1, "row_ror:4") "s_nop 0\n\t"
        EVC_DPP_ADD_(0, "row_ror:8") EVC_DPP_ADD_(1, "row_ror:8")
        : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void row_allreduce_u32x1(unsigned& a) {
    asm("s_nop 1\n\t" EVC_DPP_ADD_(0, "row_ror:1") "s_nop 1\n\t" EVC_DPP_ADD_(0, "row_ror:2") "s_nop 1\n\t"
        EVC_DPP_ADD_(0, "row_ror:4") "s_nop 1\n\t" EVC_DPP_ADD_(0, "row_ror:8")
        : "+v"(a));
}
#undef EVC_DPP_ADD_
template <int WORDS>
__device__ __forceinline__ void row_allreduce_words(unsigned (&w)[WORDS]) {
    static_assert(WORDS >= 1 && WORDS <= 8, "packed class-sum words");
    constexpr int kFull = WORDS / 3 * 3;
#pragma unroll
    for (int i = 0; i < kFull; i += 3) row_allreduce_u32x3(w[i], w[i + 1], w[i + 2]);
    if constexpr (WORDS - kFull == 2) row_allreduce_u32x2(w[kFull], w[kFull + 1]);
    if constexpr (WORDS - kFull == 1) row_allreduce_u32x1(w[kFull]);
}
// Sum over the 16 lanes of a row, IDENTICAL (bit for bit) on all of them: a butterfly — neighbours inside pairs, pairs inside
// quads (quad_perm), quads inside halves (row_half_mirror), halves (row_mirror) — adds, at every level, the two halves of
// a symmetric pair, and a + b == b + a exactly.  Rounds 1-2 used four rotations (row_ror 1, 2, 4, 8): the same cost, but
// every lane then adds the sixteen values in its own order and the results differ in the last bits between lanes.  The
// in-row water-filling branches on those sums per lane (bracket updates, the convergence test): on saturated / discrete
// actions two lanes of a row took different branches and the Newton iteration settled on a point that met the cap with a
// different multiplier per lane — found in round 3 by the replay form of the rollout tests (1 environment in 1022 x 628
// steps; `tests/test_gpu_rollout.py`, `tools/scratch/ring_step_diff.py`).
__device__ __forceinline__ double row_allreduce_f64(double v) {
    v += dpp_f64<0xB1, 0xf, true>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E, 0xf, true>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141, 0xf, true>(v);   // row_half_mirror
    v += dpp_f64<0x140, 0xf, true>(v);   // row_mirror
    return v;
}
// the same butterfly in float32 (each step is one v_add_f32 with a DPP operand)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_allreduce_f32(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    return v;
}
// does any lane of MY row have `flag` set?
__device__ __forceinline__ bool row_any(bool flag, unsigned row) {
    const unsigned long long b = __ballot(flag);
    return ((unsigned)(b >> (row * 16u)) & 0xffffu) != 0u;
}

constexpr unsigned kOob = 0xffffffffu;     // byte offset that is out of range for every buffer
constexpr unsigned kBadIdx = 0x1fffffffu;  // element index whose *4 and *8 byte offsets are out of range
                                           // (arrays are < 2 GiB, checked at evc_create)

// |M_c S|^2 (float32 screen) / |M_c S| (float64) for row c = q from per-row packed class sums
template <int WORDS>
__device__ __forceinline__ float quad_mag2_f32(const LdsNet& net, unsigned c, const unsigned (&tot)[WORDS]) {
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

// Upper bound of the projected action (amps): min(32, demand_f32 / A_PERS_TO_KWH), env.py:188-189
// with demands = previous float32 observation (env.py:218).  The quotient is a reciprocal multiply
// + one fma correction (Markstein): correctly rounded like the divide, 3 instructions.
__device__ __forceinline__ double quad_demand_cap(int dep, double rem) {
    const bool active = (dep != kEmptyDep) && (rem > Consts::FULLY_CHARGED_EPS);
    const double dm = (double)(float)rem;
    const double q0 = dm * (1.0 / Consts::A_PERS_TO_KWH);
    const double rr = fma(-q0, Consts::A_PERS_TO_KWH, dm);
    const double cap = fmin(fma(rr, 1.0 / Consts::A_PERS_TO_KWH, q0), Consts::ACTION_SCALE_FACTOR);
    return active ? cap : 0.0;
}

__device__ __forceinline__ double row_allreduce_min_f64(double v) {
    v = fmin(v, dpp_f64<0x121, 0xf, false>(v));
    v = fmin(v, dpp_f64<0x122, 0xf, false>(v));
    v = fmin(v, dpp_f64<0x124, 0xf, false>(v));
    v = fmin(v, dpp_f64<0x128, 0xf, false>(v));
    return v;
}

// Exact float64 constraint rows of schedule y for the rows flagged `want` (row-uniform): returns,
// for lane q < m, whether constraint row q is violated; cap_viol = classes above their cap.
// G / class_cap: the class count and the class caps (Params::class_cap), passed as they are held by the caller — the
// streaming kernels keep a copy in LDS: in a rarely taken branch every `P.` field is a scalar load from the kernel-argument
// segment that misses the scalar cache, one dependent round trip per loop iteration.
__device__ __forceinline__ bool quad_exact_rows(int G, const double* class_cap, const LdsNet& net, unsigned q, unsigned m,
                                                const int (&st_gid)[kSlots], const double (&y)[kSlots],
                                                bool want, unsigned& cap_viol, double tol = Consts::PROJ_TOL) {
    double re = 0.0, im = 0.0;
    cap_viol = 0u;
    for (int g0 = 0; g0 < G; g0 += 4) {            // four independent reduction ladders in flight (a lone one is all latency)
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
                if (S[u] > class_cap[g] * (1.0 + tol)) cap_viol |= 1u << g;
            }
        }
    }
    return want && q < m && sqrt(re * re + im * im) > net.mag[q] * (1.0 + tol);
}

// Water-filling of class g inside each row flagged `on` (see waterfill_class in evc_kernels.h):
// nu >= 0 with sum_{class g} clip(b - nu, 0, h) = cap; safeguarded Newton, row-local reductions.
// y is updated in place.
__device__ __forceinline__ void quad_waterfill(bool on, int g, const int (&st_gid)[kSlots],
                                               const float (&act)[kSlots], const int (&dep)[kSlots],
                                               const double (&rem)[kSlots], double cap,
                                               double (&y)[kSlots], const bool (&is_cc)[kSlots], unsigned long long* counters, int tie_log2,
                                               unsigned long long* pass_count = nullptr, bool trace = false) {
    // target b and cap h of every slot, once (slots that are compile-time empty fold away)
    double b[kSlots], h[kSlots];
    bool in_g[kSlots];
#pragma unroll
    for (int j = 0; j < kSlots; j++) {
        in_g[j] = st_gid[j] == g;
        b[j] = (double)act[j] * Consts::ACTION_SCALE_FACTOR;
        h[j] = quad_demand_cap(dep[j], rem[j]);
    }
    double nu = 0.0, lo = 0.0, hi = 64.0;
    bool run = on;
#ifdef EVC_FILL_WARM32
    // float32 passes first (same safeguarded Newton; full-rate arithmetic, one-instruction DPP adds, v_rcp_f32 instead of
    // the IEEE float64 divide) find the multiplier to ~1e-6; the float64 loop below then starts beside the solution — one
    // Newton step on the right active set lands, the next pass confirms — instead of walking there from nu = 0.  Only
    // the START of the float64 iteration changes: bracket, safeguards and stopping rule are untouched, so whatever the
    // float32 passes return the result meets the cap to 1e-13 as before.
    {
        float bf[kSlots], hf[kSlots];
#pragma unroll
        for (int j = 0; j < kSlots; j++) {
            bf[j] = act[j] * (float)Consts::ACTION_SCALE_FACTOR;
            hf[j] = (float)h[j];
        }
        const float capf = (float)cap;
        float nuf = 0.0f, lof = 0.0f, hif = 64.0f;
        bool runf = on;
        for (int it = 0; it < 10 && __ballot(runf) != 0ull; it++) {
            float part = 0.0f;
            unsigned nfree = 0u;
#pragma unroll
            for (int j = 0; j < kSlots; j++) {
                const float v = bf[j] - nuf;
                part += in_g[j] ? fminf(fmaxf(v, 0.0f), hf[j]) : 0.0f;
                nfree += (in_g[j] && v > 0.0f && v <= hf[j] && hf[j] > 0.0f) ? 1u : 0u;
            }
            const float f = row_allreduce_f32(part) - capf;
            const unsigned kfree = row_allreduce_u32(nfree);
            if (runf) {
                if (fabsf(f) <= 1e-5f * capf || kfree == 0u) {
                    runf = false;                 // close enough, or a flat piece (the float64 loop owns the breakpoints)
                } else {
                    if (f > 0.0f) lof = nuf; else hif = nuf;
                    float nxt = nuf + f * __builtin_amdgcn_rcpf((float)kfree);
                    if (!(nxt > lof && nxt < hif)) nxt = 0.5f * (lof + hif);
                    nuf = nxt;
                }
            }
        }
        if (on) nu = (double)nuf;
    }
#endif
#ifdef EVC_ABL_FILL_PASSES         /* ablation builds only (wrong results): the Newton loop cut off after so many passes */
    for (int it = 0; it < (EVC_ABL_FILL_PASSES) && __ballot(run) != 0ull; it++) {
#else
    for (int it = 0; it < 80 && __ballot(run) != 0ull; it++) {
#endif
        if (pass_count) *pass_count += 1ull;
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
        // the next breakpoint is needed only on a flat piece above the cap (every station of the class clamped at its
        // upper bound): its min-ladder runs behind a wave-uniform branch instead of in every pass
        double bp = 0.0;
        if (__builtin_expect(__ballot(run && kfree == 0u && f > 0.0) != 0ull, 0)) {
            double nextbp = 1e300;
#pragma unroll
            for (int j = 0; j < kSlots; j++) nextbp = fmin(nextbp, (in_g[j] && b[j] - nu > h[j]) ? b[j] - h[j] : 1e300);
            bp = row_allreduce_min_f64(nextbp);
        }
#ifdef EVC_TRACE_FILL
        if (trace) printf("fill g %d it %d lane %u run %d nu %.17g f %.17g kfree %u lo %.17g hi %.17g bp %.6g\n", g, it, (unsigned)__lane_id(), (int)run, nu, f, kfree, lo, hi, bp);
#endif
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
    for (int j = 0; j < kSlots; j++) {
        if (on && in_g[j]) {
            const double yw = fmin(fmax(b[j] - nu, 0.0), h[j]);
            // tie snap of solver-moved values
            y[j] = (yw != fmin(b[j], h[j])) ? tie_snap_counted(yw, h[j], is_cc[j], counters, tie_log2) : yw;
        }
    }
}

#ifndef EVC_QUAD_WAVES
#define EVC_QUAD_WAVES 4
#endif
template <bool PROJECT, int WORDS, bool DBG>
__global__ __launch_bounds__(256, EVC_QUAD_WAVES) void step_kernel_quad(Params P, StepIO io) {
    __shared__ LdsNet net;
    stage_net(net, P);
    const unsigned lane = threadIdx.x & 63u, q = lane & 15u, row = lane >> 4;
    const unsigned n = (unsigned)P.n, F = (unsigned)P.F;
    const unsigned m = (unsigned)P.m, k = (unsigned)P.k;
    const unsigned N = (unsigned)P.N;

    // ---- per-lane, per-slot station constants ----
    bool st_valid[kSlots], st_cc[kSlots];
    int st_word[kSlots], st_shift[kSlots], st_gid[kSlots];
#pragma unroll
    for (int j = 0; j < kSlots; j++) {
        const unsigned s = (unsigned)j * 16u + q;
        st_valid[j] = s < n;
        st_cc[j] = (P.cc_mask >> s) & 1ull;
        int gid = 0;
        for (int g = 0; g < P.G; g++)
            if ((P.group_mask[g] >> s) & 1ull) gid = g;
        st_word[j] = gid >> 1;
        st_shift[j] = (gid & 1) << 4;
        st_gid[j] = st_valid[j] ? gid : -1;
    }

    // ---- buffer resources over whole arrays (lane supplies a 32-bit byte offset) ----
    const rsrc_t r_rem = row_rsrc(P.rem, N * n * 8u);
    const rsrc_t r_de = row_rsrc(P.depest, N * n * 4u);
    const rsrc_t r_act = row_rsrc(io.actions, (!DBG || io.actions) ? N * n * 4u : 0u);
    const rsrc_t r_scal = row_rsrc(P.scal, N * 32u);
    const rsrc_t r_acc = row_rsrc(P.acc, N * 24u);
    const rsrc_t r_obs = row_rsrc(io.out.obs, N * F * 4u);
    const rsrc_t r_moer = row_rsrc(P.moer_obs, (unsigned)P.moer_days * EVC_MOER_ROWS * EVC_MOER_COLS * 4u);
    const rsrc_t r_hist = row_rsrc(P.moer_hist, (unsigned)P.moer_days * EVC_MOER_ROWS * 8u);
    const rsrc_t r_ts = row_rsrc(P.tables->timestep, EVC_MOER_ROWS * 4u);
    const rsrc_t r_rew = row_rsrc(io.out.reward, N * 8u);
    const rsrc_t r_term = row_rsrc(io.out.terminated, N);
    const rsrc_t r_bd = row_rsrc(io.out.breakdown, io.out.breakdown ? N * 24u : 0u);
    const rsrc_t r_sess = row_rsrc(P.sessions, (unsigned)P.bank_slots * (unsigned)P.max_sessions * 8u);
    const rsrc_t r_req = row_rsrc(P.requested, (unsigned)P.bank_slots * (unsigned)P.max_sessions * 8u);

    const bool stepwise = P.battery_stepwise != 0;
    const unsigned nquads = (N + 3u) >> 2;
    EnvWalker walk((int)nquads, 4);            // XCD-aware walk over quads
    struct QuadLoads {
        v4u s0, s1;
        double rem[kSlots];
        unsigned de[kSlots];
        float a[kSlots];
        double acc;
    };
    auto issue = [&](int quad_) {
        QuadLoads L;
        const unsigned env_ = (unsigned)quad_ * 4u + row;
        const bool ev_ = quad_ < walk.hi && env_ < N;
        const unsigned soff = ev_ ? env_ * 32u : kOob;
        L.s0 = __builtin_amdgcn_raw_buffer_load_b128(r_scal, (int)soff, 0, 0);
        L.s1 = __builtin_amdgcn_raw_buffer_load_b128(r_scal, (int)soff, 16, 0);
#pragma unroll
        for (int j = 0; j < kSlots; j++) {
            const bool v = ev_ && st_valid[j];
            const unsigned idx = v ? env_ * n + (unsigned)j * 16u + q : kBadIdx;
            L.rem[j] = buf_ld_f64(r_rem, idx * 8u);
            L.de[j] = buf_ld_u32(r_de, idx * 4u);
            const bool greedy = DBG && io.action_kind == EVC_ACTION_GREEDY;
            L.a[j] = 0.0f;
            if (!greedy) L.a[j] = buf_ld_f32(r_act, idx * 4u);
        }
        L.acc = buf_ld_f64(r_acc, (ev_ && q < 3u) ? env_ * 24u + q * 8u : kOob);
        return L;
    };
    for (int quad = walk.first; quad < walk.hi; quad += walk.stride) {
        const unsigned env = (unsigned)quad * 4u + row;
        const bool ev = env < N;
        // (a register prefetch of the next quad costs this kernel a wave of occupancy: measured slower)
        const QuadLoads cur = issue(quad);
        const v4u s0 = cur.s0, s1 = cur.s1;
        double rem[kSlots];
        int dep[kSlots], est[kSlots];
        float act[kSlots];
#pragma unroll
        for (int j = 0; j < kSlots; j++) {
            const bool v = ev && st_valid[j];
            rem[j] = cur.rem[j];
            dep[j] = v ? (int)(short)(cur.de[j] & 0xffffu) : kEmptyDep;
            est[j] = (int)cur.de[j] >> 16;
            act[j] = cur.a[j];
            if (DBG && io.action_kind == EVC_ACTION_GREEDY)     // baselines.py:32-35 on the observation
                act[j] = (dep[j] != kEmptyDep && rem[j] > Consts::FULLY_CHARGED_EPS) ? 1.0f : 0.0f;
        }
        const double acc_in = cur.acc;

        int t = (int)s0.x, cursor = (int)s0.y, slot = (int)s0.z, moer_day = (int)s0.w;
        int n_sessions = (int)s1.x, next_arrival = (int)s1.y, status = (int)s1.z, episodes = (int)s1.w;
        const bool after_done = ev && t >= EVC_EPISODE_STEPS;   // step() after termination w/o autoreset
        bool live = ev && !after_done;
        const int t1 = t + 1;

        // MOER loads for t1 (row-uniform addresses)
        const unsigned mrow = live ? ((unsigned)moer_day * EVC_MOER_ROWS + (unsigned)t1) : 0u;
        const double moer_now = buf_ld_f64(r_hist, live ? mrow * 8u : kOob);
        float mo[3];
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const unsigned idx = (unsigned)p * 16u + q;                 // position in [forecast | prev | ts]
            const unsigned col = idx < k ? idx + 1u : 0u;
            mo[p] = buf_ld_f32(r_moer, (live && idx <= k) ? (mrow * EVC_MOER_COLS + col) * 4u : kOob);
            if (idx == k + 1u) mo[p] = buf_ld_f32(r_ts, live ? (unsigned)t1 * 4u : kOob);
        }

        // ---- action -> y (box clip of the projection) ----
        bool clamped = false;
        double y[kSlots];
        unsigned ywords[WORDS], pwords[WORDS];
#pragma unroll
        for (int w = 0; w < WORDS; w++) { ywords[w] = 0u; pwords[w] = 0u; }
#pragma unroll
        for (int j = 0; j < kSlots; j++) {
            float a = act[j];
            clamped = clamped || (st_valid[j] && !(a >= 0.0f && a <= 1.0f));
            a = fminf(fmaxf(a, 0.0f), 1.0f);
            act[j] = a;
            double yy = (double)a * Consts::ACTION_SCALE_FACTOR;        // env.py:366
            if (PROJECT) {
                yy = fmin(yy, quad_demand_cap(dep[j], rem[j]));
                const unsigned qy = (unsigned)(int)ceil(yy * 8.0) << st_shift[j];
#pragma unroll
                for (int w = 0; w < WORDS; w++) ywords[w] += (st_word[j] == w) ? qy : 0u;
            }
            y[j] = yy;
        }

        // ---- projection screen (PROJECT) ----
        bool pilots_screened = false;
        if (PROJECT) {
#pragma unroll
            for (int w = 0; w < WORDS; w++) ywords[w] = row_allreduce_u32(ywords[w]);
            bool maybe = false, maybe_p = false;
            if (q < m) {
                const float mag2 = quad_mag2_f32<WORDS>(net, q, ywords);
                maybe = !(mag2 < net.thr_y2[q]);
                maybe_p = !(mag2 < net.thr_yp2[q]);
            }
            bool undecided = live && row_any(maybe, row);
            pilots_screened = !row_any(maybe_p, row);
            if (__ballot(undecided) != 0ull) {
                // Rare (wave-uniform branch): exact float64 rows for the undecided rows; class-cap
                // (pod breaker) violations are projected in closed form by water-filling inside the
                // row; anything else is queued for the slow kernel.
                unsigned cap_viol;
                bool hard = quad_exact_rows(P.G, P.class_cap, net, q, m, st_gid, y, undecided, cap_viol);
                bool anyviol = row_any(hard, row);
                // Class caps are filled whenever one is violated, also beside violated multi-class rows:
                // if the point projected onto box and caps satisfies every row it is the projection
                // (relaxation argument); what remains violated goes to the slow kernel.
                bool fill = undecided && cap_viol != 0u;
                if (__ballot(fill) != 0ull) {
                    // y is updated in place: rows that cannot be settled here are queued and the
                    // slow kernel recomputes them from the stored state
                    for (int g = 0; g < P.G; g++) {
                        const bool do_g = fill && ((cap_viol >> g) & 1u);
                        if (__ballot(do_g) != 0ull) quad_waterfill(do_g, g, st_gid, act, dep, rem, P.class_cap[g], y, st_cc, DBG ? P.tie_counters : nullptr, P.tie_log2);
                    }
                    unsigned cv2;
                    // re-verify every row on the snapped values: Params::snap_tol = PROJ_TOL + the most the snap can add to a row
                    const bool still = row_any(quad_exact_rows(P.G, P.class_cap, net, q, m, st_gid, y, fill, cv2, P.snap_tol), row);
                    anyviol = anyviol && !(fill && !still);
                }
                const bool queue_me = undecided && anyviol;       // cones (or unsettled): slow kernel
                if (queue_me && q == 0u) queue_push(P, (int)env);
                live = live && !queue_me;                 // queued rows write nothing here
                pilots_screened = pilots_screened && !undecided;
            }
        }

        // ---- pilots (env.py:366-378), battery charge, class sums of the pilots ----
        double amps_sum = 0.0;
        double pilot[kSlots], amps[kSlots];      // kept only for the DBG outputs
#pragma unroll
        for (int j = 0; j < kSlots; j++) {
            const double pl = st_valid[j] ? legal_pilot(y[j], st_cc[j]) : 0.0;
            pilot[j] = pl;
            const unsigned qp = (unsigned)(int)pl << st_shift[j];
#pragma unroll
            for (int w = 0; w < WORDS; w++) pwords[w] += (st_word[j] == w) ? qp : 0u;
            const bool occupied = dep[j] != kEmptyDep;
            amps[j] = charge_ev(occupied ? pl : 0.0, rem[j], stepwise);
            amps_sum += amps[j];
        }

        // ---- reductions inside the row ----
        const double total_rate = row_allreduce_f64(amps_sum);           // env.py:445
#pragma unroll
        for (int w = 0; w < WORDS; w++) pwords[w] = row_allreduce_u32(pwords[w]);
        double excess = 0.0;
        {
            bool maybe = false;
            if (q < m && !pilots_screened) maybe = !(quad_mag2_f32<WORDS>(net, q, pwords) < net.thr_p2[q]);
            if (__ballot(maybe && live) != 0ull) {                       // rare: exact evaluation
                double ex = 0.0;
                if (q < m) ex = fmax(row_mag_f64<WORDS>(net, (int)q, pwords, 1.0) - net.mag[q], 0.0);
                excess = row_allreduce_f64(ex);
            }
        }

        // ---- acnsim event pass at iteration t1: unplug (precedence 0) before plug-in (10) ----
#pragma unroll
        for (int j = 0; j < kSlots; j++)
            if (dep[j] != kEmptyDep && dep[j] <= t1) { dep[j] = kEmptyDep; est[j] = 0; rem[j] = 0.0; }
        bool pending = live && next_arrival <= t1 && cursor < n_sessions;
        while (__ballot(pending) != 0ull) {
            const unsigned sidx = (unsigned)slot * (unsigned)P.max_sessions + (unsigned)cursor;
            const v2u sv = __builtin_amdgcn_raw_buffer_load_b64(r_sess, pending ? (int)(sidx * 8u) : (int)kOob, 0, 0);
            const double rq = buf_ld_f64(r_req, pending ? sidx * 8u : kOob);
            const int s_dep = (int)(short)(sv.x >> 16);
            const int s_est = (int)(short)(sv.y & 0xffffu);
            const unsigned s_st = sv.y >> 16;
            const bool mine = pending && (s_st & 15u) == q;
            const unsigned sj = s_st >> 4;
            bool busy = false;
#pragma unroll
            for (int j = 0; j < kSlots; j++) busy = busy || (mine && sj == (unsigned)j && dep[j] != kEmptyDep);
            const bool row_busy = row_any(busy, row);
            if (pending && row_busy) status |= EVC_STATUS_OCCUPIED;      // acnportal: StationOccupiedError
#pragma unroll
            for (int j = 0; j < kSlots; j++)
                if (mine && !row_busy && sj == (unsigned)j) { dep[j] = s_dep; est[j] = s_est; rem[j] = rq; }
            if (pending) {
                cursor += 1;
                next_arrival = kNoArrival;
            }
            const bool more = pending && cursor < n_sessions;
            const unsigned nx = buf_ld_u32(r_sess, more ? (sidx + 1u) * 8u : kOob);
            if (more) next_arrival = (int)(short)(nx & 0xffffu);
            pending = more && next_arrival <= t1;
        }
        if (live) t = t1;
        if (live && row_any(clamped, row)) status |= EVC_STATUS_ACTION_CLAMPED;
        const bool done = live && t1 >= EVC_EPISODE_STEPS;
        if (after_done) status |= EVC_STATUS_STEP_AFTER_DONE;

        // ---- reward (env.py:431-464) ----
        const double profit = Consts::PROFIT_FACTOR * total_rate;
        const double carbon = Consts::CARBON_COST_FACTOR * total_rate * moer_now;
        const double excess_charge = excess * Consts::VIOLATION_FACTOR;
        const double reward = after_done ? 0.0 : profit - carbon - excess_charge;
        const double acc = acc_in + ((q == 0u) ? profit : (q == 1u ? carbon : excess_charge));
        const bool wr = live || after_done;           // rows that report reward / terminated
        buf_st_f64(r_rew, (wr && q == 0u) ? env * 8u : kOob, reward);
        if (DBG && io.out.returns && live && q == 0u) io.out.returns[env] += reward;
        buf_st_u8(r_term, (wr && q == 0u) ? env : kOob, (done || after_done) ? 1 : 0);
        buf_st_f64(r_bd, (live && q < 3u) ? env * 24u + q * 8u : kOob, acc);
        if (DBG) {                                                        // debug / parity outputs
#pragma unroll
            for (int j = 0; j < kSlots; j++) {
                const size_t idx = (size_t)env * n + (unsigned)j * 16u + q;
                if (live && st_valid[j]) {
                    if (io.out.pilots) io.out.pilots[idx] = pilot[j];
                    if (io.out.rates) io.out.rates[idx] = amps[j];
                    if (io.out.projected) io.out.projected[idx] = y[j] / Consts::ACTION_SCALE_FACTOR;
                }
            }
        }

        // ---- autoreset (gymnasium VectorEnv): terminal observation, then next episode's state ----
        const bool do_reset = done && P.autoreset;
        if (done) episodes += 1;
        if (__ballot(do_reset) != 0ull) {
            if (io.out.final_obs) {
                const rsrc_t r_fin = row_rsrc(io.out.final_obs, N * F * 4u);
#pragma unroll
                for (int j = 0; j < kSlots; j++) {
                    const bool active = (dep[j] != kEmptyDep) && (rem[j] > Consts::FULLY_CHARGED_EPS);
                    const unsigned o = (do_reset && st_valid[j]) ? (env * F + (unsigned)j * 16u + q) * 4u : kOob;
                    buf_st_f32(r_fin, o, active ? (float)rem[j] : 0.0f);
                    buf_st_f32(r_fin, o == kOob ? kOob : o + n * 4u, active ? (float)(est[j] - t) : 0.0f);
                }
#pragma unroll
                for (int p = 0; p < 3; p++) {
                    const unsigned idx = (unsigned)p * 16u + q;
                    buf_st_f32(r_fin, (do_reset && idx < k + 2u) ? (env * F + 2u * n + idx) * 4u : kOob, mo[p]);
                }
            }
            if (do_reset) {
                const int next = (slot + P.autoreset_stride) % P.bank_slots;
                slot = next;
                t = 0; cursor = 0;
                moer_day = P.slot_moer_day[next];
                n_sessions = P.n_sessions[next];
                next_arrival = n_sessions > 0 ? (int)P.sessions[(size_t)next * P.max_sessions].arrival : kNoArrival;
#pragma unroll
                for (int j = 0; j < kSlots; j++) { rem[j] = 0.0; dep[j] = kEmptyDep; est[j] = 0; }
            }
            const unsigned mrow0 = (unsigned)moer_day * EVC_MOER_ROWS;
#pragma unroll
            for (int p = 0; p < 3; p++) {
                const unsigned idx = (unsigned)p * 16u + q;
                const unsigned col = idx < k ? idx + 1u : 0u;
                const float v = buf_ld_f32(r_moer, (do_reset && idx <= k) ? (mrow0 * EVC_MOER_COLS + col) * 4u : kOob);
                if (do_reset) mo[p] = (idx == k + 1u) ? 0.0f : v;
            }
        }

        // ---- observation (env.py:381-394) + state write-back ----
#pragma unroll
        for (int j = 0; j < kSlots; j++) {
            const bool active = (dep[j] != kEmptyDep) && (rem[j] > Consts::FULLY_CHARGED_EPS);
            const bool wv = live && st_valid[j];
            const unsigned sidx = wv ? env * n + (unsigned)j * 16u + q : kBadIdx;
            const unsigned oidx = wv ? env * F + (unsigned)j * 16u + q : kBadIdx;
            buf_st_f32(r_obs, oidx * 4u, active ? (float)rem[j] : 0.0f);
            buf_st_f32(r_obs, (oidx + n) * 4u, active ? (float)(est[j] - t) : 0.0f);
            buf_st_f64(r_rem, sidx * 8u, rem[j]);
            buf_st_u32(r_de, sidx * 4u, (unsigned)((dep[j] & 0xffff) | (est[j] << 16)));
        }
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const unsigned idx = (unsigned)p * 16u + q;
            buf_st_f32(r_obs, (live && idx < k + 2u) ? (env * F + 2u * n + idx) * 4u : kOob, mo[p]);
        }
        buf_st_f64(r_acc, (live && q < 3u) ? env * 24u + q * 8u : kOob, do_reset ? 0.0 : acc);
        {
            v4u o0, o1;
            o0.x = (unsigned)t; o0.y = (unsigned)cursor; o0.z = (unsigned)slot; o0.w = (unsigned)moer_day;
            o1.x = (unsigned)n_sessions; o1.y = (unsigned)next_arrival; o1.z = (unsigned)status; o1.w = (unsigned)episodes;
            const unsigned so = ((live || after_done) && q == 0u) ? env * 32u : kOob;
            __builtin_amdgcn_raw_buffer_store_b128(o0, r_scal, (int)so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(o1, r_scal, (int)so, 16, 0);
        }
    }
}

}  // namespace evc
