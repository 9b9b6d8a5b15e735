// evc_rollout.h — the fused multi-period rollout: T consecutive EVChargingEnv.step() calls under a
// device-resident policy in ONE launch (the reference's episode loop is
// BaseAlgorithm.run, algorithms/base.py:63-88, over GreedyAlgorithm / RandomAlgorithm,
// algorithms/evcharging/baselines.py:22-51).
//
// Geometry of evc_cquad.h — a wavefront = four 16-lane DPP rows = four environments, lane q of a row
// owns entry slots q, q+16, q+32, q+48 of its environment — but the wavefront KEEPS its quad for all T
// periods: the plugged-in EVs (entry word + remaining demand), the event cursor, the reward accumulators
// and the episode return live in registers from the first period to the last.  Per period there is
//   * no action read: GreedyAlgorithm's action is `remaining demand > 0` of the entry itself;
//     RandomAlgorithm's is Philox4x32-10 of (seed, env, episode, period, station), drawn by lane q of the
//     row for stations 4q..4q+3 — the stream of random_actions_kernel / evc_fill_random_actions, bit for bit;
//   * no observation write, no state write-back: the observation a policy of this kind needs IS the state;
//   * one row-uniform load (the MOER value of the period) and the session records that arrive;
//   * no launch boundary: wavefronts are never synchronised with each other, so a wavefront that meets a
//     congested period (exact rows, water-filling, an iterative solve) only delays its own four
//     environments — there is no per-period tail for the whole batch to wait for.
// An environment whose projection needs the iterative solver (cone rows binding) is solved where it stands:
// its row hands targets and caps to a station-shaped LDS image, the wavefront runs the slow path's
// solve_projection (evc_solver.h) behind one real call — same relaxation sequence, same tie snap as the
// slow kernel — and the entries read their projected values back.
// After the last period the state goes back to memory in the compact layout and the outputs of the LAST
// step (observation, reward, terminated, breakdown; returns += every reward) are written once: the launch is
// equivalent to T calls of evc_step (terminal observations of episodes that end inside the rollout included).
#pragma once

#include "evc_cquad.h"
#include "evc_rowcone.h"
#include "evc_rollout_launch.h"

namespace evc {

constexpr unsigned kEmptyMeta = 0xffffffffu;      // entry word of a free slot (departure 1023: never reached)

struct RolloutLds {
    LdsNet net;
    SolverWs ws[4];                                // one solver workspace per wavefront
    float act_img[4][4][64];                       // [wave][row][station] action of this period (random policy)
    // [wave][row][station], zero at rest, two users: (a) float2 {demand, est_departure} while an observation row is
    // written; (b) the hand-off of the iterative projection: the row's entries put their caps h there, the solve leaves
    // the projected values y in the same cells
    union Cell {
        float2 obs;
        double h_or_y;
    } img[4][4][64];
    uint4 st_mulw[64];
    uint4 st_mulw_hi[64];
    unsigned char st_info[64];
    LdsRare rare;                                  // Params fields of the rarely taken projection branch
    int noconv[4];                                 // per wavefront: bit r = the solve of row r did not converge
    // warm start of the general iteration (greedy policy; evc_solver.h SolveWarm): the multipliers it ended with for the row's
    // environment, and the period (loop count + 1) they belong to — used only in the very next period
    double zwarm[4][4][16][2];
    int warm_tag[4][4];
    int capv[4][4];                                // hand-off: the classes the period's body has filled for the row (0: none)
};
static_assert(sizeof(SolverWs) >= 4 * 64 * sizeof(double), "a wavefront's solver workspace doubles as the hand-off image of filled values");
static_assert(EVC_MAX_CONSTRAINTS >= 16, "the quad geometry handles up to 16 rows");

constexpr unsigned kRolloutKernargBytes = (unsigned)((((sizeof(Params) + alignof(RolloutIO) - 1) / alignof(RolloutIO)) * alignof(RolloutIO) + sizeof(RolloutIO) + 7u) & ~(size_t)7u);

// The iterative projection of the rows in `rows` (bit r = row r of wavefront wv; bit 4 = random policy): lane i =
// station i of the row's environment.  Caps h come from S.img[wv][r] (0 where no EV is plugged in), targets are
// b = 32 a with a = the action image (random policy) or 1 (greedy: a = 1 wherever the cap is non-zero, and where the
// cap is zero the target does not matter); the projected values go back into the same cells.
// Behind a real call for the reason given at drain_local_list (evc_cquad.h): inlined, the solver's live ranges
// would sit in the period loop.  Params come from the kernel-argument segment.
__device__ __attribute__((noinline)) void rollout_solve_rows(unsigned lds, unsigned wv, unsigned rows) {
    const char* ka = (const char*)__builtin_amdgcn_implicitarg_ptr() - kRolloutKernargBytes;
    const Params& P = *(const Params*)ka;
    typedef __attribute__((address_space(3))) RolloutLds LdsImage;
    RolloutLds& S = *(RolloutLds*)(LdsImage*)(size_t)(unsigned)rfl((int)lds);
    const int w = rfl((int)wv);
    unsigned mask = (unsigned)rfl((int)rows);
    const unsigned tag = mask >> 8;                // 0: no warm start (policies whose actions are redrawn every period)
    mask &= 0xffu;
    const int lane = (int)__lane_id();
#ifndef EVC_NO_ROWCONE             /* variant builds only: the round-3 form (every row through the wave-per-environment solver) */
    {
        // Round 4: all flagged rows at once, each in its own 16-lane row (evc_rowcone.h): box clip -> caps -> one cone row -> two.
        // What that settles is written back and taken out of `mask`; the general path below only sees what is left.
        const unsigned q = (unsigned)lane & 15u, row = (unsigned)lane >> 4;
        const bool on = ((mask >> row) & 1u) != 0u;
        int st_gid[kSlots];
        bool is_cc[kSlots];
        double b[kSlots], h[kSlots], y[kSlots], ywin[kSlots];
        // the schedule the period's body arrived at (box clip + its fillings), station-shaped, in the memory of this wavefront's
        // solver workspace (free until the general path below needs it)
        const double* ywimg = reinterpret_cast<const double*>(&S.ws[w]) + row * 64u;
#pragma unroll
        for (int j = 0; j < kSlots; j++) {
            const unsigned st = (unsigned)j * 16u + q;
            const bool valid = st < (unsigned)P.n;
            const unsigned info = S.st_info[st];
            st_gid[j] = valid ? (int)(info & 0x7fu) : -1;
            is_cc[j] = (info >> 7) != 0u;
            ywin[j] = valid ? ywimg[st] : 0.0;
            h[j] = valid ? S.img[w][row][st].h_or_y : 0.0;
            b[j] = !valid ? 0.0 : (mask & 16u) ? (double)S.act_img[w][row][st] * Consts::ACTION_SCALE_FACTOR : Consts::ACTION_SCALE_FACTOR;
        }
        const bool warm_on = tag != 0u && P.m <= 16;
        const bool warm_ok = warm_on && tag > 1u && S.warm_tag[w][row] == (int)tag - 1;
        bool stored = false;
        const bool settled = quad_project(S.rare.G, S.rare.class_cap, S.rare.simple_rows, P.tie_counters, S.rare.tie_log2, S.net, q, (unsigned)P.m, row, on, st_gid, is_cc, b, h, y,
                                          warm_on ? S.zwarm[w][row] : nullptr, warm_ok, &stored,
                                          ywin, on ? (unsigned)S.capv[w][row] : 0u, S.rare.snap_tol);
        if (warm_on && on && settled && q == 0u) S.warm_tag[w][row] = stored ? (int)tag : 0;
        if (settled) {
#pragma unroll
            for (int j = 0; j < kSlots; j++)
                if (st_gid[j] >= 0) S.img[w][row][(unsigned)j * 16u + q].h_or_y = y[j];
        }
        const unsigned long long sb = __ballot(settled);
        const unsigned done = ((sb & 0xffffull) ? 1u : 0u) | ((sb & 0xffff0000ull) ? 2u : 0u) | ((sb & 0xffff00000000ull) ? 4u : 0u) |
                              ((sb >> 48) ? 8u : 0u);
        mask &= ~(unsigned)rfl((int)done);
        SOLVER_SYNC();
        if ((mask & 15u) == 0u) return;
#if defined(EVC_RABL) && EVC_RABL == 5   /* ablation (WRONG results): rows the row-form solver did not settle are left as they are */
        return;
#endif
    }
#endif
    SolverLds L(S.net, S.ws[w]);
    const LaneNet lnet = lane_net(P, lane);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        if (!((mask >> r) & 1u)) continue;
        SolverLane ln;
        ln.gid = lnet.gid;
        ln.h = S.img[w][r][lane].h_or_y;
        ln.b = (mask & 16u) ? (double)S.act_img[w][r][lane] * Consts::ACTION_SCALE_FACTOR : Consts::ACTION_SCALE_FACTOR;
        ln.y = 0.0;
        bool noconv;
        double y;
        if (tag != 0u && P.m <= 16) {
            SolveWarm W{S.zwarm[w][r], S.warm_tag[w][r] == (int)tag - 1 && tag > 1u, false};
            y = solve_projection(P, L, lnet, ln, lane, noconv, &W);
            if (lane == 0) S.warm_tag[w][r] = W.stored ? (int)tag : 0;
        } else {
            y = solve_projection(P, L, lnet, ln, lane, noconv);
        }
        S.img[w][r][lane].h_or_y = y;
        if (noconv && lane == 0) S.noconv[w] |= 1 << r;
        SOLVER_SYNC();
    }
}

#ifndef EVC_ROLLOUT_WAVES
#define EVC_ROLLOUT_WAVES 3
#endif

// KIND: 0 = greedy, 1 = random (Philox), 2 = replay of a pre-staged action ring (evc_rollout with EVC_ACTION_F32 / _DISCRETE)
// WAVES: wavefronts per SIMD the register allocation is held to.  3 (168 VGPRs) leaves the projecting copies ~570 spilled
// VGPRs, all of them around the rare projection branch: free on quiet days (4.1 / 5.0e9 env-steps/s greedy / random against
// 3.2 / 4.0e9 at 2), but on the reference's GMM days at Caltech three quads in four take that branch at midday and the scratch
// traffic is what the wavefronts wait for (SQ_WAIT_ANY 327 of 650 wave cycles): 2 (256 VGPRs, ~90 spilled) runs those 2.40
// against 1.95e9.  JPL's GMM days prefer 3 again (2.36 against 2.16e9).  The engine therefore MEASURES (evc_engine.hip,
// launch_rollout): the projecting kernels exist at both settings and the faster one on the caller's own workload is kept.
// NC / ALIVE: as step_kernel_cquad's (evc_cquad.h) — the site's shape compiled in, and every environment known to be inside its
// episode (whole quads, autoreset, clocks in range): the engine launches these copies when both hold, the general form otherwise.
// Per period the predicates `ev` / `live` / `after_done` fold: synthetic greedy 13.4 -> 11.2 us per period, GMM greedy 28.9 -> 25.7
// (profiles/r6_rollassume_ab.txt).
template <bool PROJECT, int WORDS, int KIND, int WAVES = EVC_ROLLOUT_WAVES, int NC = 0, bool ALIVE = false>
__global__ __launch_bounds__(256, WAVES) void rollout_kernel(Params P, RolloutIO io) {
    static_assert(!ALIVE || NC != 0, "ALIVE comes with a compiled-in shape");
    __shared__ RolloutLds S;
    LdsNet& net = S.net;
    auto& st_mulw = S.st_mulw;
    auto& st_mulw_hi = S.st_mulw_hi;
    auto& st_info = S.st_info;

    const unsigned tid = threadIdx.x, lane = tid & 63u, q = lane & 15u, row = lane >> 4;
    const unsigned wv = (unsigned)rfl((int)(tid >> 6));
    const unsigned n = NC ? (unsigned)NC : (unsigned)P.n, k = NC ? (unsigned)kSiteForecast : (unsigned)P.k;
    const unsigned F = NC ? 2u * n + k + 2u : (unsigned)P.F;
    const unsigned m = (unsigned)P.m;
    const unsigned N = (unsigned)P.N;

    // ---- workgroup prologue: network tables, per-station multipliers, clean images ----
    if (tid < 64u) {
        const unsigned s = tid;
        const bool valid = s < n;
        int gid = 0;
        for (int g = 0; g < P.G; g++)
            if ((P.group_mask[g] >> s) & 1ull) gid = g;
        unsigned mw[8];
#pragma unroll
        for (int w = 0; w < 8; w++) mw[w] = (valid && (gid >> 1) == w) ? ((gid & 1) ? 65536u : 1u) : 0u;
        st_mulw[s] = make_uint4(mw[0], mw[1], mw[2], mw[3]);
        st_mulw_hi[s] = make_uint4(mw[4], mw[5], mw[6], mw[7]);
        st_info[s] = (unsigned char)((unsigned)gid | ((unsigned)((P.cc_mask >> s) & 1ull) << 7));
    }
    if (tid < 4u) S.noconv[tid] = 0;
    if (tid < 16u) S.warm_tag[tid >> 2][tid & 3u] = 0;
#pragma unroll
    for (int j = 0; j < kSlots; j++) S.img[wv][row][j * 16 + q].h_or_y = 0.0;
    stage_rare(S.rare, P);
    stage_net(net, P);                              // ends with the workgroup barrier

    const unsigned nquads = (N + 3u) >> 2;
    // persistent wavefronts (round 6): the first quad by position, every further one from RolloutIO::quad_counter
    // (positions in RolloutIO::quad_order, busiest quads first, when the engine supplies it)
    auto quad_at = [&](unsigned pos) {
        unsigned qd = pos;
        if (io.quad_order != nullptr && pos < nquads) qd = (unsigned)rfl((int)io.quad_order[pos]);
        return pos < nquads ? qd : nquads;
    };
    auto next_quad = [&]() {
        unsigned nq = 0u;
        if (lane == 0u) nq = __hip_atomic_fetch_add(io.quad_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return quad_at((unsigned)rfl((int)nq));
    };
    for (unsigned quad = quad_at(blockIdx.x * 4u + wv); quad < nquads; quad = next_quad()) {      // (after the only barrier)
    // what the previous quad of this wavefront left in its share of the workgroup's LDS: warm-start tags are periods of THAT quad
    if (q == 0u) S.warm_tag[wv][row] = 0;
    if (lane == 0u) S.noconv[wv] = 0;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const unsigned env = quad * 4u + row;
    const bool ev = ALIVE || env < N;
    const unsigned ebase = env * n;

    const rsrc_t r_win = row_rsrc(P.win_base, P.win_span);
    const Win r_rem{r_win, P.off_rem}, r_de{r_win, P.off_de}, r_scal{r_win, P.off_scal}, r_acc{r_win, P.off_acc};
    const Win r_sess{r_win, P.off_sess}, r_req{r_win, P.off_req}, r_hist{r_win, P.off_hist};

    float* const act_row = S.act_img[wv][row];
    RolloutLds::Cell* const img_row = S.img[wv][row];
    auto lds_sync = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    };
    auto station_mulw = [&](unsigned st, unsigned (&mw)[WORDS]) {
        const uint4 lo = st_mulw[st];
        const unsigned all[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
        for (int w = 0; w < WORDS && w < 4; w++) mw[w] = all[w];
        if (WORDS > 4) {
            const uint4 hi = st_mulw_hi[st];
            const unsigned allh[4] = {hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int w = 4; w < WORDS; w++) mw[w] = allh[w - 4];
        }
    };

    // ---- the quad's state: memory -> registers ----
    int t, cursor, slot, moer_day, n_sessions, next_arrival, status, episodes;
    unsigned meta[kSlots];
    double rem[kSlots];
    {
        const unsigned soff = ev ? env * 32u : kOob;
        const v4u s0 = buf_ld_v4(r_scal, soff);
        const v4u s1 = buf_ld_v4(r_scal, soff == kOob ? kOob : soff + 16u);
        t = (int)s0.x; cursor = (int)s0.y; slot = (int)s0.z; moer_day = (int)s0.w;
        n_sessions = (int)s1.x; next_arrival = (int)s1.y; status = (int)s1.z & kStatusMask; episodes = (int)s1.w;
        const unsigned A = ev ? ((s1.z >> kCountShift) & 0x7fu) : 0u;
#pragma unroll
        for (int c = 0; c < kSlots; c++) {
            const unsigned e = (unsigned)c * 16u + q;
            const unsigned w = buf_ld_u32(r_de, e < A ? (ebase + e) * 4u : kOob);
            const double r = buf_ld_f64(r_rem, e < A ? (ebase + e) * 8u : kOob);
            meta[c] = e < A ? w : kEmptyMeta;
            rem[c] = e < A ? r : 0.0;
        }
        if (!ev) { t = EVC_EPISODE_STEPS; next_arrival = kNoArrival; n_sessions = 0; cursor = 0; }
    }
    double acc = buf_ld_f64(r_acc, (ev && q < 3u) ? env * 24u + q * 8u : kOob);   // lane q < 3: accumulator q
    double bd = acc;                                 // breakdown as of the last live step
    double ret = (io.out.returns && ev) ? io.out.returns[env] : 0.0;
    double reward = 0.0;
    bool done_last = false, ever_live = false;

    const bool stepwise = P.battery_stepwise != 0;
    const unsigned blocks_per_env = (n + 3u) >> 2;

    // observation of the row's CURRENT state (env.py:381-394) into `dst` (obs / final_obs) for rows with `on`
    auto emit_obs = [&](float* dst_base, bool on) {
        const rsrc_t r_dst = row_rsrc(dst_base, N * F * 4u);
        const unsigned obase = env * F;
#pragma unroll
        for (int c = 0; c < kSlots; c++) {
            if (on && meta[c] != kEmptyMeta) {
                const bool active = rem[c] > Consts::FULLY_CHARGED_EPS;
                img_row[entry_station(meta[c])].obs = make_float2(active ? (float)rem[c] : 0.0f,
                                                                  active ? (float)(entry_est(meta[c]) - t) : 0.0f);
            }
        }
        lds_sync();
#pragma unroll
        for (int j = 0; j < kSlots; j++) {
            const float2 d = img_row[j * 16 + q].obs;
            img_row[j * 16 + q].h_or_y = 0.0;
            const bool w = on && (unsigned)j * 16u + q < n;
            const unsigned o = (obase + (unsigned)j * 16u + q) * 4u;
            buf_st_f32(r_dst, w ? o : kOob, d.x);
            buf_st_f32(r_dst, w ? o + n * 4u : kOob, d.y);
        }
        const unsigned mrow = (unsigned)moer_day * EVC_MOER_ROWS + (unsigned)t;
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const unsigned idx = (unsigned)p * 16u + q;                 // position in [forecast | prev | ts]
            const unsigned col = idx < k ? idx + 1u : 0u;
            const unsigned o_moer = P.off_moer + (mrow * EVC_MOER_COLS + col) * 4u;
            const unsigned o_ts = P.off_ts + (unsigned)t * 4u;
            const float v = buf_ld_f32(r_win, (on && idx <= k + 1u) ? (idx <= k ? o_moer : o_ts) : kOob);
            buf_st_f32(r_dst, (on && idx < k + 2u) ? (obase + 2u * n + idx) * 4u : kOob, v);
        }
        lds_sync();
    };

    constexpr bool RANDOM = KIND != 0;                            // actions come through the action image (drawn or replayed)
    // replay: lane q of a row fetches stations 4q..4q+3 of its environment's action row, one period ahead
    const bool discrete_in = KIND == 2 && io.policy == EVC_ACTION_DISCRETE;
    const unsigned act_elem = discrete_in ? 8u : 4u;
    const rsrc_t r_ring = row_rsrc(KIND == 2 ? io.actions : nullptr, KIND == 2 ? (unsigned)io.ring_len * N * n * act_elem : 0u);
    auto fetch_actions = [&](int step_) {
        v4u raw;
        const unsigned base = ((unsigned)(step_ % io.ring_len) * N + env) * n;
        const unsigned w[4] = {0u, 0u, 0u, 0u};
        (void)w;
        unsigned r4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned s_ = 4u * q + (unsigned)j;
            r4[j] = buf_ld_u32(r_ring, (KIND == 2 && ev && step_ < io.steps && s_ < n) ? (base + s_) * act_elem : kOob);
        }
        raw.x = r4[0]; raw.y = r4[1]; raw.z = r4[2]; raw.w = r4[3];
        return raw;
    };
    v4u act_next = fetch_actions(0);

#ifdef EVC_ROLLOUT_STATS      /* measurement builds (tools/rollout_stats.py): per wavefront into Params::slow_list — 100 MHz ticks of the whole loop, inside the rare projection branch, inside the solve call; visits and calls */
    unsigned ms_t0 = (unsigned)__builtin_amdgcn_s_memrealtime(), ms_rare = 0u, ms_call = 0u, ms_calls = 0u, ms_visits = 0u, ms_exact = 0u, ms_fill = 0u, ms_short = 0u, ms_fills = 0u, ms_fillcalls = 0u;
    unsigned long long ms_passes = 0ull;
#endif
    for (int step = 0; step < io.steps; step++) {
        const bool after_done = !ALIVE && ev && t >= EVC_EPISODE_STEPS;   // step() after termination w/o autoreset
        const bool live = ev && !after_done;
        if (after_done) { status |= EVC_STATUS_STEP_AFTER_DONE; reward = 0.0; done_last = true; }
        if (__ballot(live) == 0ull) break;                        // every later step of this quad is the same no-op
        ever_live = ever_live || live;
        const int t1 = t + 1;
        // loads of the period, issued first: the MOER value of the reward, the session record at the cursor
        const double moer_now = buf_ld_f64(r_hist, live ? ((unsigned)moer_day * EVC_MOER_ROWS + (unsigned)t1) * 8u : kOob);
        bool pending = live && next_arrival <= t1 && cursor < n_sessions;
        unsigned sidx = (unsigned)slot * (unsigned)P.max_sessions + (unsigned)cursor;
        v2u sv = buf_ld_v2(r_sess, pending ? sidx * 8u : kOob);
        double rq = buf_ld_f64(r_req, pending ? sidx * 8u : kOob);

        if (KIND == 2) {                                        // replayed actions: clamp like evc_step (NaN -> 0, flag)
            const v4u raw = act_next;
            act_next = fetch_actions(step + 1);
            const unsigned rw[4] = {raw.x, raw.y, raw.z, raw.w};
            float a[4];
            bool clamped = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float in = discrete_in ? (float)(int)rw[j] / (float)(io.bins - 1) : __uint_as_float(rw[j]);   // wrappers.py:43-45
                a[j] = fminf(fmaxf(in, 0.0f), 1.0f);
                clamped = clamped || (4u * q + (unsigned)j < n && a[j] != in);
            }
            if (live && row_any(clamped, row)) status |= EVC_STATUS_ACTION_CLAMPED;
            *reinterpret_cast<float4*>(&act_row[4u * q]) = make_float4(a[0], a[1], a[2], a[3]);
            lds_sync();
        }
        // action image of the random policy: lane q draws stations 4q..4q+3 of its row
        if (KIND == 1) {
            float4 a4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (live && q < blocks_per_env) {
                const Philox ph((unsigned)t | (q << 16), (unsigned)episodes, io.env_id_base + env, kPolicyTag,
                                (unsigned)io.seed, (unsigned)(io.seed >> 32));
                float a[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned w = ph.w[j];
                    if (io.bins >= 2) a[j] = (float)(unsigned)(((unsigned long long)w * (unsigned)io.bins) >> 32) / (float)(io.bins - 1);
                    else a[j] = (float)(w >> 8) * (1.0f / 16777216.0f);
                }
                a4 = make_float4(a[0], a[1], a[2], a[3]);
            }
            *reinterpret_cast<float4*>(&act_row[4u * q]) = a4;
            lds_sync();
        }
        const bool station_pilots = !PROJECT && RANDOM;          // see evc_cquad.h
        double total_rate = 0.0, excess = 0.0;                   // what the period's body hands to the reward

        auto body = [&](auto ns_tag) {
        constexpr int NS = decltype(ns_tag)::value;
        // ---- entries: decode, action, y (box clip of the projection), class sums ----
        bool valid[kSlots];
        unsigned st[kSlots];
        int dep[kSlots];
        float act[kSlots];
        double y[kSlots];
        unsigned ywords[WORDS], pwords[WORDS];
#pragma unroll
        for (int w = 0; w < WORDS; w++) { ywords[w] = 0u; pwords[w] = 0u; }
#pragma unroll
        for (int c = 0; c < kSlots; c++) { valid[c] = false; st[c] = 0u; dep[c] = kEmptyDep; act[c] = 0.0f; y[c] = 0.0; }
#pragma unroll
        for (int c = 0; c < NS; c++) {
            valid[c] = live && meta[c] != kEmptyMeta;
            st[c] = (unsigned)entry_station(meta[c]);
            dep[c] = valid[c] ? entry_dep(meta[c]) : kEmptyDep;
            float a = RANDOM ? act_row[st[c]] : ((rem[c] > Consts::FULLY_CHARGED_EPS) ? 1.0f : 0.0f);   // baselines.py:32-35 / :45-51
            a = valid[c] ? a : 0.0f;
            act[c] = a;
            double yy = (double)a * Consts::ACTION_SCALE_FACTOR;        // env.py:366
            if (PROJECT) {
                yy = fmin(yy, quad_demand_cap(dep[c], rem[c]));
                const unsigned qy = (unsigned)(int)ceil(yy * 8.0);
                unsigned mw[WORDS];
                station_mulw(st[c], mw);
#pragma unroll
                for (int w = 0; w < WORDS; w++) ywords[w] = __umul24(qy, mw[w]) + ywords[w];
            }
            y[c] = yy;
        }

        // ---- projection (env.py:178-221): screen -> exact rows -> in-row water-filling -> iterative solve ----
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
            const bool undecided = live && row_any(maybe, row);
            pilots_screened = !row_any(maybe_p, row);
#if defined(EVC_RABL) && EVC_RABL == 1   /* ablation builds only (WRONG results): the period without the rare projection branch */
            if (false) {
#else
            if (__builtin_expect(__ballot(undecided) != 0ull, 0)) {
#endif
#ifdef EVC_ROLLOUT_STATS
                const unsigned ms_r0 = (unsigned)__builtin_amdgcn_s_memrealtime();
                ms_visits += 1u;
#endif
                int st_gid[kSlots];
#pragma unroll
                for (int c = 0; c < kSlots; c++) st_gid[c] = valid[c] ? (int)(st_info[st[c]] & 0x7fu) : -1;
                unsigned cap_viol;
                bool anyviol;
                // Caps-only shortcut (this kernel only; measured and not adopted in step_kernel_cquad):
                // where the screen left only simple rows (class caps) open on a network whose rows are monotone in every
                // class sum (Params::monotone_rows), capping those classes IS the projection and no row is evaluated again.
                // Why the second quad_exact_rows check of the step kernels can be dropped here: (i) a capped class ends at
                // its cap up to the water-filling's tolerance plus the tie snap, at most n_g * 2^-17 A — exactly the slack
                // Params::snap_tol is derived from, so that check passes by construction; (ii) every other row was cleared
                // by the screen at the box-clipped point on class sums of ceil(8 y) / 8, the capping only lowers class sums
                // (monotone rows stay cleared), and the snap moves a value by < 2^-16 A where the screen's thresholds keep
                // the float32 error of a whole row in hand.  The equivalence with a loop of evc_step calls is tested with
                // the shortcut on and off (tests/test_gpu_rollout.py::test_caps_shortcut_on_and_off...).
                const unsigned open_rows = (unsigned)(__ballot(maybe) >> (row * 16u)) & 0xffffu;
                const bool caps_only = S.rare.monotone_rows != 0 && (open_rows & ~S.rare.simple_rows) == 0u;
                const bool shortcut = __ballot(undecided && !caps_only) == 0ull;
                if (shortcut) {
                    cap_viol = 0u;
                    const int G_ = S.rare.G;
                    const unsigned capped = S.rare.cap_classes;
                    for (int g = 0; g < G_; g++) {
                        if (!((capped >> g) & 1u)) continue;
                        double part = 0.0;
#pragma unroll
                        for (int c = 0; c < kSlots; c++) part += (st_gid[c] == g) ? y[c] : 0.0;
                        if (row_allreduce_f64(part) > S.rare.class_cap[g] * (1.0 + Consts::PROJ_TOL)) cap_viol |= 1u << g;
                    }
                    anyviol = false;
                } else {
                    const bool hard = quad_exact_rows(S.rare.G, S.rare.class_cap, net, q, m, st_gid, y, undecided, cap_viol);
                    anyviol = row_any(hard, row);
                }
#ifdef EVC_ROLLOUT_STATS
                const unsigned ms_r1 = (unsigned)__builtin_amdgcn_s_memrealtime();
                ms_exact += ms_r1 - ms_r0;
                ms_short += shortcut ? 1u : 0u;
#endif
#if defined(EVC_RABL) && EVC_RABL == 2   /* ablation (WRONG results): exact rows only */
                const bool fill = false;
                anyviol = false;
#else
                const bool fill = undecided && cap_viol != 0u;
#endif
                if (__builtin_expect(__ballot(fill) != 0ull, 0)) {
                    bool slot_cc[kSlots];
#pragma unroll
                    for (int c = 0; c < kSlots; c++) slot_cc[c] = valid[c] && (st_info[st[c]] >> 7) != 0u;
                    for (int g = 0; g < S.rare.G; g++) {
                        const bool do_g = fill && ((cap_viol >> g) & 1u);
#ifdef EVC_ROLLOUT_STATS
                        if (__ballot(do_g) != 0ull) { ms_fillcalls += 1u; quad_waterfill(do_g, g, st_gid, act, dep, rem, S.rare.class_cap[g], y, slot_cc, nullptr, S.rare.tie_log2, &ms_passes); }
#else
                        if (__ballot(do_g) != 0ull) quad_waterfill(do_g, g, st_gid, act, dep, rem, S.rare.class_cap[g], y, slot_cc, nullptr, S.rare.tie_log2);
#endif
                    }
#if defined(EVC_RABL) && EVC_RABL == 3   /* ablation (WRONG results): exact rows + filling, no second evaluation, no solve */
                    anyviol = false;
                    if (false) {
#else
                    if (!shortcut) {
#endif
                        unsigned cv2;
                        const bool still = row_any(quad_exact_rows(S.rare.G, S.rare.class_cap, net, q, m, st_gid, y, fill, cv2, S.rare.snap_tol), row);
                        anyviol = anyviol && !(fill && !still);
                    }
                }
#if defined(EVC_RABL) && EVC_RABL == 4   /* ablation (WRONG results): everything but the solve */
                anyviol = false;
#endif
#ifdef EVC_ROLLOUT_STATS
                ms_fill += (unsigned)__builtin_amdgcn_s_memrealtime() - ms_r1;
                ms_fills += __ballot(fill) != 0ull ? 1u : 0u;
#endif
                const bool solve_me = undecided && anyviol;       // cone rows bind (or unsettled): iterative solver
                const unsigned long long solve_mask = __ballot(solve_me);
                if (__builtin_expect(solve_mask != 0ull, 0)) {
                    // the rows' caps, station-shaped (free stations stay 0)
#pragma unroll
                    for (int c = 0; c < NS; c++)
                        if (solve_me && valid[c]) img_row[st[c]].h_or_y = quad_demand_cap(dep[c], rem[c]);
                    // ... and what the filling above made of them (round 5: the row-form solver does not fill again)
                    double* const ywrow = reinterpret_cast<double*>(&S.ws[wv]) + row * 64u;
#pragma unroll
                    for (int j = 0; j < kSlots; j++) ywrow[j * 16 + q] = 0.0;
                    if (q == 0u) S.capv[wv][row] = (solve_me && fill) ? (int)cap_viol : 0;
                    lds_sync();
#pragma unroll
                    for (int c = 0; c < NS; c++)
                        if (solve_me && valid[c]) ywrow[st[c]] = y[c];
                    const unsigned rows = ((solve_mask & 0xffffull) ? 1u : 0u) | ((solve_mask & 0xffff0000ull) ? 2u : 0u) |
                                          ((solve_mask & 0xffff00000000ull) ? 4u : 0u) | ((solve_mask >> 48) ? 8u : 0u) |
                                          (KIND != 0 ? 16u : 0u) | (KIND == 0 ? (unsigned)(step + 1) << 8 : 0u);
                    lds_sync();
#ifdef EVC_ROLLOUT_STATS
                    const unsigned ms_c0 = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
                    rollout_solve_rows((unsigned)(size_t)(__attribute__((address_space(3))) void*)&S, wv, rows);
#ifdef EVC_ROLLOUT_STATS
                    ms_call += (unsigned)__builtin_amdgcn_s_memrealtime() - ms_c0; ms_calls += 1u;
#endif
                    lds_sync();
#pragma unroll
                    for (int c = 0; c < NS; c++)
                        if (solve_me && valid[c]) {
                            y[c] = img_row[st[c]].h_or_y;
                            img_row[st[c]].h_or_y = 0.0;
                        }
                    if (solve_me && ((S.noconv[wv] >> row) & 1)) status |= EVC_STATUS_PROJ_NOCONV;
                    lds_sync();
                    if (lane == 0u) S.noconv[wv] = 0;
                    lds_sync();
                }
                pilots_screened = pilots_screened && !undecided;
#ifdef EVC_ROLLOUT_STATS
                ms_rare += (unsigned)__builtin_amdgcn_s_memrealtime() - ms_r0;
#endif
            }
        }

#ifdef EVC_ROLLOUT_TRACE          /* debug builds only: -DEVC_ROLLOUT_TRACE=<env> -DEVC_ROLLOUT_TRACE_T=<t> */
        if (ev && env == (unsigned)(EVC_ROLLOUT_TRACE) && t == (EVC_ROLLOUT_TRACE_T)) {
#pragma unroll
            for (int c = 0; c < NS; c++)
                if (valid[c]) printf("trace NS=%d lane %u slot %d st %u dep %d rem %.9f act %.3f y %.9f screened %d\n", NS, lane, c, st[c], dep[c], rem[c], (double)act[c], y[c], (int)pilots_screened);
        }
#endif
        // ---- pilots (env.py:366-378), battery charge ----
        double pilot[kSlots], amps[kSlots], rem_in[kSlots];
        bool cross[kSlots];
#pragma unroll
        for (int c = 0; c < kSlots; c++) { pilot[c] = 0.0; amps[c] = 0.0; rem_in[c] = 0.0; cross[c] = false; }
#pragma unroll
        for (int c = 0; c < NS; c++) {
            const unsigned info = st_info[st[c]];
            const double pl = legal_pilot(y[c], (info >> 7) != 0u);     // y = 0 on free slots
            pilot[c] = pl;
            rem_in[c] = rem[c];
            amps[c] = charge_ev_main(pl, rem[c], stepwise, cross[c]);    // a free slot (or a row that is not live) has pilot 0: nothing moves
        }
        {
            bool any_cross = false;
#pragma unroll
            for (int c = 0; c < NS; c++) any_cross = any_cross || cross[c];
            if (__builtin_expect(__ballot(any_cross) != 0ull, 0)) {
#pragma unroll
                for (int c = 0; c < NS; c++)
                    if (cross[c]) amps[c] = charge_ev_cross(pilot[c], rem_in[c], rem[c]);
            }
        }
        double amps_sum = 0.0;
#pragma unroll
        for (int c = 0; c < NS; c++) amps_sum += amps[c];
        total_rate = row_allreduce_f64(amps_sum);                        // env.py:445

        // ---- constraint excess of the pilots (env.py:449-452) ----
        if (station_pilots) {
#pragma unroll
            for (int j = 0; j < kSlots; j++) {
                const unsigned s = (unsigned)j * 16u + q;
                const unsigned info = st_info[s];
                const float a = act_row[s];
                const double ps = (live && s < n) ? legal_pilot((double)a * Consts::ACTION_SCALE_FACTOR, (info >> 7) != 0u) : 0.0;
                unsigned mw[WORDS];
                station_mulw(s, mw);
#pragma unroll
                for (int w = 0; w < WORDS; w++) pwords[w] = __umul24((unsigned)(int)ps, mw[w]) + pwords[w];
            }
        }
        if (station_pilots || __ballot(live && !pilots_screened) != 0ull) {
            if (!station_pilots) {
#pragma unroll
                for (int c = 0; c < NS; c++) {
                    unsigned mw[WORDS];
                    station_mulw(st[c], mw);
#pragma unroll
                    for (int w = 0; w < WORDS; w++) pwords[w] = __umul24((unsigned)(int)pilot[c], mw[w]) + pwords[w];
                }
            }
#pragma unroll
            for (int w = 0; w < WORDS; w++) pwords[w] = row_allreduce_u32(pwords[w]);
            bool maybe = false;
            if (q < m && !pilots_screened) maybe = !(quad_mag2_f32<WORDS>(net, q, pwords) < net.thr_p2[q]);
            if (__builtin_expect(__ballot(maybe && live) != 0ull, 0)) {   // rare: exact evaluation
                double ex = 0.0;
                if (q < m) ex = fmax(row_mag_f64<WORDS>(net, (int)q, pwords, 1.0) - net.mag[q], 0.0);
                excess = row_allreduce_f64(ex);
            }
        }

        // ---- acnsim event pass at iteration t1: unplug (precedence 0) before plug-in (10) ----
#pragma unroll
        for (int c = 0; c < NS; c++)
            if (valid[c] && dep[c] <= t1) { meta[c] = kEmptyMeta; rem[c] = 0.0; }
        if (NS > 1) {                                   // entries sink to the lowest free slot of their lane
#pragma unroll
            for (int c = 1; c < NS; c++)
                if (meta[c - 1] == kEmptyMeta && meta[c] != kEmptyMeta) {
                    meta[c - 1] = meta[c]; rem[c - 1] = rem[c];
                    meta[c] = kEmptyMeta; rem[c] = 0.0;
                }
        }
        };
        // the widest occupied slot of the wavefront picks the copy of the period's body
        if (__builtin_expect(__ballot(meta[1] != kEmptyMeta || meta[2] != kEmptyMeta || meta[3] != kEmptyMeta) != 0ull, 0)) {
            if (__ballot(meta[3] != kEmptyMeta) != 0ull) body(std::integral_constant<int, 4>{});
            else if (__ballot(meta[2] != kEmptyMeta) != 0ull) body(std::integral_constant<int, 3>{});
            else body(std::integral_constant<int, 2>{});
        } else {
            body(std::integral_constant<int, 1>{});
        }

        // ---- plug-ins of iteration t1 (sessions are sorted by arrival; the record at the cursor was fetched above) ----
        if (live) t = t1;
        while (__ballot(pending) != 0ull) {
            const int s_dep = (int)(short)(sv.x >> 16);
            const int s_est = (int)(short)(sv.y & 0xffffu);
            const unsigned s_st = (sv.y >> 16) & 63u;
            bool busy = false;
#pragma unroll
            for (int c = 0; c < kSlots; c++)
                busy = busy || (pending && meta[c] != kEmptyMeta && (unsigned)entry_station(meta[c]) == s_st);
            const bool row_busy = row_any(busy, row);
            if (pending && row_busy) status |= EVC_STATUS_OCCUPIED;      // acnportal: StationOccupiedError
            // the new entry takes the lowest free slot of the lowest free lane of its row (n <= 64 = 16 lanes x 4 slots:
            // there is always one)
            bool placed = !(pending && !row_busy);
            const unsigned new_meta = pack_entry(s_dep, (int)s_st, s_est);
#pragma unroll
            for (int c = 0; c < kSlots; c++) {
                if (__ballot(!placed) != 0ull) {
                    const unsigned freebits = (unsigned)(__ballot(meta[c] == kEmptyMeta) >> (row * 16u)) & 0xffffu;
                    const bool here = !placed && freebits != 0u;
                    const unsigned tgt = (unsigned)__builtin_ctz(freebits | 0x10000u);
                    if (here && q == tgt) { meta[c] = new_meta; rem[c] = rq; }
                    placed = placed || here;
                }
            }
            if (pending) {
                cursor += 1;
                sidx += 1u;
                next_arrival = kNoArrival;
            }
            const bool more_ev = pending && cursor < n_sessions;
            sv = buf_ld_v2(r_sess, more_ev ? sidx * 8u : kOob);
            rq = buf_ld_f64(r_req, more_ev ? sidx * 8u : kOob);
            if (more_ev) next_arrival = (int)(short)(sv.x & 0xffffu);
            pending = more_ev && next_arrival <= t1;
        }

        // ---- reward (env.py:431-464), accumulators, episode return ----
        const double profit = Consts::PROFIT_FACTOR * total_rate;
        const double carbon = Consts::CARBON_COST_FACTOR * total_rate * moer_now;
        const double excess_charge = excess * Consts::VIOLATION_FACTOR;
        const bool done = live && t1 >= EVC_EPISODE_STEPS;
        if (live) {
            reward = profit - carbon - excess_charge;
            ret += reward;
            acc = acc + ((q == 0u) ? profit : (q == 1u ? carbon : excess_charge));
            bd = acc;
            done_last = done;
        }
        if (done) episodes += 1;

        // ---- autoreset (gymnasium VectorEnv): terminal observation, then the next episode of the bank ----
        const bool do_reset = done && P.autoreset;
        if (__builtin_expect(__ballot(do_reset) != 0ull, 0)) {
            if (io.out.final_obs) emit_obs(io.out.final_obs, do_reset);
            if (do_reset) {
                const int next = (slot + P.autoreset_stride) % P.bank_slots;
                slot = next;
                t = 0; cursor = 0;
                const int first_arrival = (int)P.sessions[(size_t)next * P.max_sessions].arrival;
                moer_day = P.slot_moer_day[next];
                n_sessions = P.n_sessions[next];
                next_arrival = n_sessions > 0 ? first_arrival : kNoArrival;
#pragma unroll
                for (int c = 0; c < kSlots; c++) { meta[c] = kEmptyMeta; rem[c] = 0.0; }
                acc = 0.0;
            }
        }
    }

#ifdef EVC_ROLLOUT_STATS
    if (lane == 0u && quad < 16384u) {
        unsigned* o = (unsigned*)P.slow_list + quad * 4u;
        o[0] = (unsigned)__builtin_amdgcn_s_memrealtime() - ms_t0; o[1] = ms_rare; o[2] = ms_call; o[3] = ms_calls | (ms_visits << 16);
        // (the list holds N ints = four per wavefront: the second half of the wavefronts reports the split of the visit instead)
        // second record: exact-rows ticks, filling ticks, visits that took the caps-only shortcut, visits with a filling
        if (quad >= 8192u && quad < 16384u) { o[0] = (unsigned)ms_passes | (ms_fillcalls << 16); o[1] = ms_exact; o[2] = ms_fill; o[3] = ms_short | (ms_fills << 16); }
    }
#endif
    // ---- registers -> memory: compact state, then the outputs of the last step ----
    {
        unsigned count = 0u;
#pragma unroll
        for (int c = 0; c < kSlots; c++) {
            const bool alive = meta[c] != kEmptyMeta;
            const unsigned bits = (unsigned)(__ballot(alive) >> (row * 16u)) & 0xffffu;
            const unsigned pos = count + (unsigned)__popc(bits & ((1u << q) - 1u));
            count += (unsigned)__popc(bits);
            const bool w = ev && alive;
            buf_st_u32(r_de, w ? (ebase + pos) * 4u : kOob, meta[c]);
            buf_st_f64(r_rem, w ? (ebase + pos) * 8u : kOob, rem[c]);
        }
        buf_st_f64(r_acc, (ev && q < 3u) ? env * 24u + q * 8u : kOob, acc);
        v4u o0, o1;
        o0.x = (unsigned)t; o0.y = (unsigned)cursor; o0.z = (unsigned)slot; o0.w = (unsigned)moer_day;
        o1.x = (unsigned)n_sessions; o1.y = (unsigned)next_arrival;
        o1.z = ((unsigned)status & (unsigned)kStatusMask) | (count << kCountShift);
        o1.w = (unsigned)episodes;
        const unsigned so = (ev && q == 0u) ? env * 32u : kOob;
        buf_st_v4(r_scal, so, o0);
        buf_st_v4(r_scal, so == kOob ? kOob : so + 16u, o1);
    }
    if (io.steps > 0) {
        if (__ballot(ever_live) != 0ull) emit_obs(io.out.obs, ever_live);
        const rsrc_t r_rew = row_rsrc(io.out.reward, N * 8u);
        const rsrc_t r_term = row_rsrc(io.out.terminated, N);
        const rsrc_t r_bd = row_rsrc(io.out.breakdown, io.out.breakdown ? N * 24u : 0u);
        buf_st_f64(r_rew, (ev && q == 0u) ? env * 8u : kOob, reward);
        buf_st_u8(r_term, (ev && q == 0u) ? env : kOob, done_last ? 1 : 0);
        buf_st_f64(r_bd, (ever_live && q < 3u) ? env * 24u + q * 8u : kOob, bd);
        if (io.out.returns && ev && q == 0u) io.out.returns[env] = ret;
    }
    }   // next quad of this wavefront
}

}  // namespace evc
