// evc_cquad.h — the streaming step kernel on the COMPACT state layout (Params::compact = 1).
//
// Same wavefront geometry as evc_quad.h (4 DPP rows of 16 lanes, row r = environment 4*quad + r) but
// the per-station arithmetic — action clip, demand cap, pilot rounding, battery charge, class sums —
// runs over the environment's ENTRIES (the EVs plugged in right now), not over its 54-64 stations:
// lane q of a row owns entries q, q+16, q+32, q+48 ("entry slots" c = 0..3).  The iteration body exists
// for 1, 2, 3 and 4 live entry slots and the widest row of the wavefront picks the copy: one slot (<= 16
// EVs) on a quiet network and at night, two or three around midday of a real Caltech / JPL day.  State
// traffic shrinks from 2 x 12 n bytes to 12 x 16 bytes read plus 12 A written.  What stays station-shaped is the interface: the action row is read densely (range
// check) and exchanged to the entries through LDS; demands / est_departures of the observation are
// scattered by the entries into a per-row LDS image and written out densely.
//   * round 6 — the station side moves in 16-BYTE accesses: lane q owns stations 4q .. 4q+3 of the action row (one
//     buffer_load_dwordx4 + one ds_write_b128 instead of four of each); the observation image is kept in the ORDER OF THE
//     OBSERVATION ROW ([demands n | est_departures n], contiguous), so lane q reads whole 16-byte chunks q and q + 16 of it
//     (ds_read_b128) and stores them as they are (buffer_store_dwordx4), and the row's tail [forecasted_moer k | prev_moer |
//     timestep] comes as ONE 16-byte load per lane from a table the engine keeps in that order (Params::off_mtail) and leaves as
//     one store: 4 stores per lane and row instead of 11, 2 loads instead of 7 (Caltech, n = 54, k = 36).  The delivered amps
//     are summed over the entries (as the fused rollout kernel does), not through the image.
//   * unplug = an entry with departure <= t+1 is simply not written back; survivors are packed by a
//     ballot prefix count; a plug-in event appends one entry (written straight to memory and to the
//     observation image by lane 0 of the row).  Entries never move between lanes.
//   * the projection screen, the in-row water-filling and the slow queue are those of evc_quad.h,
//     fed with the entries' class ids.
#pragma once

#include <cstddef>
#include <type_traits>

#include "evc_quad.h"
#include "evc_solver.h"

#if defined(EVC_TIMELINE) && EVC_TIMELINE == 3   /* measurement builds: twelve stamps inside a wavefront's SECOND quad (slots 2 .. 13) */
#define EVC_TLP(k) do { if (tl_it == 1) tl_stamp(2 + (k)); } while (0)
#else
#define EVC_TLP(k) do { } while (0)
#endif

namespace evc {

constexpr int kImgFloats = 128;          // observation image of one environment: [demands n | est_departures n], 2 n <= 128 floats

#ifndef EVC_CQUAD_WAVES
#define EVC_CQUAD_WAVES 4
#endif

// LDS of one workgroup.  `net` comes first and the per-step images follow it so that, once the workgroup's
// streaming work is done, the same memory serves as the slow path's SolverLds = {LdsNet net; workspace}
// (in-kernel queue drain, DRAIN = true).
constexpr int kDrainListMax = 256;
struct CquadLds {
    LdsNet net;
    union Images {
        struct {
            // [wave][row] image the entries scatter into and the station lanes read back in 16-byte chunks: the first 2 n
            // floats of the observation row, in its order (demand of station s at [s], est_departure at [n + s]); zero
            // between steps
            alignas(16) float obs_img[4][4][kImgFloats];
            alignas(16) float act_img[4][4][64];    // [wave][row][station] clamped action of this step
        } s;
        SolverWs solver_workspace;
    } u;
    uint4 st_mulw[64];          // per station: 0 / 1 / 65536 multipliers of packed words 0..3
    uint4 st_mulw_hi[64];       // words 4..7
    unsigned char st_info[64];  // class id | ClipperCreek << 7
    int next_quad;              // next of the workgroup's quads nobody has taken yet (see the loop over quads)
    int local_count;            // DRAIN: environments this workgroup queued for its own slow path ...
    int local_list[kDrainListMax];   // ... (the engine enables DRAIN only while a workgroup steps at most that many environments)
};

// The in-kernel drain behind a real call (DRAIN kernels): inlined, the slow path's code and live ranges cost
// the streaming path 5 us per step; with explicit arguments the caller keeps them alive (and spilled) through
// the whole streaming loop, 3 us (both measured).  So the callee takes ONE constant — the LDS address of the
// workgroup's CquadLds — and fetches everything else itself: Params / StepIO straight from the kernel-argument
// segment (uniform scalar loads; both kernels share the signature (Params, StepIO)), the list from LDS.
constexpr unsigned kStepIOKernargOffset = (unsigned)((sizeof(Params) + alignof(StepIO) - 1) / alignof(StepIO) * alignof(StepIO));
constexpr unsigned kExplicitKernargBytes = (kStepIOKernargOffset + (unsigned)sizeof(StepIO) + 7u) & ~7u;
template <int WORDS>
__device__ __attribute__((noinline)) void drain_local_list(unsigned lds) {
    // In a callable function only the IMPLICIT-argument pointer is defined: it follows the explicit kernel
    // arguments (8-byte aligned), so the explicit ones sit right below it.
    const char* ka = (const char*)__builtin_amdgcn_implicitarg_ptr() - kExplicitKernargBytes;
    const Params& P = *(const Params*)ka;
    const StepIO& io = *(const StepIO*)(ka + kStepIOKernargOffset);
    typedef __attribute__((address_space(3))) CquadLds LdsImage;
    CquadLds& S = *(CquadLds*)(LdsImage*)(size_t)(unsigned)rfl((int)lds);
    const int lane = (int)__lane_id();            // wave 0 of the workgroup: lane = thread id, without asking the caller for it
    const int count = rfl(S.local_count < kDrainListMax ? S.local_count : kDrainListMax);
    // the slow path works on the workgroup's tables and on the memory of the per-step images (CquadLds::Images)
    SolverLds L(S.net, S.u.solver_workspace);
    for (int i = 0; i < count; i++) solve_env<WORDS>(P, io, L, lane, rfl(S.local_list[i]));
}

// DRAIN: no slow kernel is launched after this one.  Each workgroup keeps the environments whose projection
// needs the iterative solver in a list of its own (LDS) and, once its streaming work is done, solves them itself
// (wave 0, one after the other, on the LDS the per-step images no longer need).  No grid-wide hand-off, no
// global queue: a dependent — almost always empty — launch costs ~2 us per step, a grid-wide "last workgroup
// drains" scheme 4 us (its ticket needs every store of the launch acknowledged first; both measured), this
// one s_barrier.  On congested steps the solves of different workgroups run side by side like the slow
// kernel's; only several queued environments inside ONE workgroup serialise (the engine falls back to the
// slow kernel once a step queues more than a few dozen, evc_engine.hip "drain mode").
//
// WAVES: wavefronts per SIMD the register allocation is held to.  At 4 (128 VGPRs) the one-slot copy of the iteration
// body fits and the wide copies (17 - 64 entries per environment) spill; at 3 (168 VGPRs) a projecting kernel has no
// spilled VGPR at all.  On the quiet benchmark day the two are equal with projection (26.3 us per step either way,
// interleaved A/B) and 4 is 1.2 us faster without; on congested days, where most wavefronts run a wide copy, 3 wins:
// Caltech's GMM day 42.1 -> 38.8 us per step, JPL's streaming kernel 45.2 -> 34.6 (38.3 at 2).  The engine launches the
// projecting lean kernels at 3 (EVC_PROJ_WAVES, evc_engine.hip), everything else at EVC_CQUAD_WAVES.
//
// GREEDY (lean projecting kernels, round 6): the device-resident GreedyAlgorithm compiled in — a run-time flag in the kernel
// that reads action rows cost the headline 0.5 - 0.7 us per step (measured, profiles/r6_lean_greedy_ab.txt), so the rule has
// instantiations of its own; the debug kernels keep testing StepIO::action_kind.
//
// NC (lean kernels, round 6): the network's shape as compile-time constants — NC stations, forecast horizon 36, hence the observation
// width 2 NC + 38 and the MOER tail's 40 floats — for the two sites the reference ships (Caltech 54, JPL 52; the engine picks the
// instantiation when Params says exactly that, any other shape runs NC = 0 = everything from Params).  Row offsets become shifts and
// adds instead of 32-bit multiplies, the chunk counts of the observation row and the branches on their remainders fold, compares
// take inline constants instead of SGPRs: VALU 84.6 -> 79.8, SALU 39.2 -> 34.7 per env-step, -2.7 % step period (profiles/r6_ab_fixn.txt).
// ALIVE (with NC): every environment of every quad exists and none stands behind the end of its episode — the batch is a whole number
// of quads, autoreset is on (an environment enters a step with t <= 287) and nobody has written other clocks into the scalars (the
// engine tracks that: evc_set_env_scalars).  `ev`, `live`, `after_done` are constants then, and with them a good part of the
// predicates folded into buffer offsets: VALU 79.8 -> 72.2, SALU 34.7 -> 30.4 per env-step, -2.5 % step period (profiles/r6_ab_live.txt).
template <bool PROJECT, int WORDS, bool DBG, bool DRAIN = false, int WAVES = EVC_CQUAD_WAVES, bool GREEDY = false, int NC = 0, bool ALIVE = false>
__global__ __launch_bounds__(256, DBG ? 2 : WAVES) void step_kernel_cquad(Params P, StepIO io) {
    static_assert(NC == 0 || !DBG, "the shape-specialised copies exist for the lean kernels");
    static_assert(!ALIVE || NC != 0, "ALIVE comes with a compiled-in shape");
    static_assert(!GREEDY || (PROJECT && !DBG), "the compiled-in greedy rule exists for the lean projecting kernels");
    static_assert(!DRAIN || (PROJECT && !DBG), "the in-kernel drain exists for the lean projecting kernel only");
    __shared__ CquadLds S;
    __shared__ double dbg_img[DBG ? 4 : 1][4][64];     // unused (and dropped) in the lean kernels
    LdsNet& net = S.net;
    auto& st_mulw = S.st_mulw;
    auto& st_mulw_hi = S.st_mulw_hi;
    auto& st_info = S.st_info;
    auto& obs_img = S.u.s.obs_img;
    auto& act_img = S.u.s.act_img;

    const unsigned tid = threadIdx.x, lane = tid & 63u, q = lane & 15u, row = lane >> 4, wv = tid >> 6;
    const unsigned n = NC ? (unsigned)NC : (unsigned)P.n, k = NC ? (unsigned)kSiteForecast : (unsigned)P.k;
    const unsigned F = NC ? 2u * n + k + 2u : (unsigned)P.F;
    const unsigned m = (unsigned)P.m;
    const unsigned N = (unsigned)P.N;

    // dense side: lane q owns stations 4q + j, j = 0..3 (action range check, act image, per-station debug outputs)
    const unsigned st4 = q * 4u;
    bool st_valid[kSlots];
#pragma unroll
    for (int j = 0; j < kSlots; j++) st_valid[j] = st4 + (unsigned)j < n;
    // observation row as 16-byte chunks: [demands | est_departures] = 2 n floats = c_full whole chunks (+ a 2-float one if n is
    // odd); tail [forecasted_moer k | prev_moer | timestep] = k + 2 floats = t_full whole chunks + t_rem floats
    const unsigned c_full = (2u * n) >> 2, t_full = (k + 2u) >> 2, t_rem = (k + 2u) & 3u;
    const unsigned mtail_w = NC ? ((k + 2u) + 3u) & ~3u : (unsigned)P.mtail_w;
    const unsigned t_chunks = mtail_w >> 2;
    const unsigned qrow = q < m ? q : 0u;             // the constraint row this lane screens

    const rsrc_t r_win = row_rsrc(P.win_base, P.win_span);      // every engine-owned array (struct Win)
    const Win r_rem{r_win, P.off_rem}, r_de{r_win, P.off_de};
    const rsrc_t r_act = row_rsrc(io.actions, io.actions ? N * n * 4u : 0u);
    const Win r_scal{r_win, P.off_scal}, r_acc{r_win, P.off_acc};
    const rsrc_t r_obs = row_rsrc(io.out.obs, N * F * 4u);
    const Win r_mtail{r_win, P.off_mtail}, r_hist{r_win, P.off_hist};
    const rsrc_t r_rew = row_rsrc(io.out.reward, N * 8u);
    const rsrc_t r_term = row_rsrc(io.out.terminated, N);
    const rsrc_t r_bd = row_rsrc(io.out.breakdown, io.out.breakdown ? N * 24u : 0u);
    const Win r_sess{r_win, P.off_sess}, r_req{r_win, P.off_req};

    // device-resident GreedyAlgorithm (baselines.py:22-35): every plugged-in EV with demand left asks for the full rate; no action
    // row is read (EVChargingVectorEnv.step(policy='greedy'))
    const bool greedy = DBG ? io.action_kind == EVC_ACTION_GREEDY : GREEDY;
    const bool stepwise = P.battery_stepwise != 0;
    const unsigned nquads = (N + 3u) >> 2;
    // this launch's share of the batch (StepIO::quad_lo / quad_hi; the whole batch unless the engine pipelines two halves)
    const int q_lo = io.quad_hi > 0 ? io.quad_lo : 0;
    EnvWalker walk(io.quad_hi > 0 ? io.quad_hi - io.quad_lo : (int)nquads, 4);
    walk.first += q_lo;
    walk.hi += q_lo;
    // The workgroup's quads {walk.first - wv + w + r * stride : w < 4, r} are shared by its four wavefronts: each takes
    // its own first one, then whichever comes next (an LDS counter).  A wavefront that ran into one of the rare
    // branches — exact rows, water-filling, a queue push: cold code and scratch reloads, 16 - 30 us per hit measured
    // on JPL's GMM afternoons, where 0.3 % of the environments took the launch from 40 to 70 us — then takes fewer
    // quads and its siblings the rest.
    const int wg_first = walk.first - (int)wv;
    auto wg_quad = [&](int k) { const int qd = wg_first + (k & 3) + (k >> 2) * walk.stride; return qd < walk.hi ? qd : -1; };
    auto take_quad = [&]() {
        int k = 0;
        if (lane == 0u) k = __hip_atomic_fetch_add(&S.next_quad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return wg_quad(rfl(k));
    };
    // Raw loads of one quad, issued one iteration ahead (software prefetch): the waves of this kernel
    // spend most of their time waiting for these round trips (SQ_WAIT_ANY = 60 % of the wave cycles).
    struct QuadRaw {
        v4u s0, s1;
        unsigned meta0;
        double rem0;
        v4u a4;                     // actions of stations 4q .. 4q+3
    };
    auto issue = [&](int quad_) {
        QuadRaw L;
        const unsigned env_ = (unsigned)quad_ * 4u + row;
        const bool ev_ = quad_ >= 0 && (ALIVE || env_ < N);
        const unsigned eb_ = env_ * n;
        const unsigned soff = ev_ ? env_ * 32u : kOob;
        L.s0 = buf_ld_v4(r_scal, soff);
        L.s1 = buf_ld_v4(r_scal, soff == kOob ? kOob : soff + 16u);
        // entry slot 0 unconditionally: its extent is known only after the scalars arrive, and a
        // dependent load would cost a second round trip
        L.meta0 = buf_ld_u32(r_de, ev_ ? (eb_ + q) * 4u : kOob);
        L.rem0 = buf_ld_f64(r_rem, ev_ ? (eb_ + q) * 8u : kOob);
        // one 16-byte load (rows are n * 4 bytes apart: 4- or 8-byte aligned, which buffer accesses allow; the range check is
        // per dword, tools/probes/b128_probe.hip); a lane whose chunk runs past station n - 1 reads the head of the next row
        // there (zeros behind the last row) and masks it with st_valid
        // (greedy: no action buffer — r_act has no records, every lane reads 0; no branch around the load)
        L.a4 = buf_ld_v4(r_act, (ev_ && st_valid[0]) ? (eb_ + st4) * 4u : kOob);
        return L;
    };
    // The first quad's loads are issued before the LDS tables are built: the two latency chains (network
    // tables from global memory, first state / action rows) overlap instead of following each other —
    // a wave runs only four iterations at N = 65 536, so the prologue is a visible share of the launch.
#ifdef EVC_TIMELINE     /* measurement builds (tools/wg_timeline.py): 16 time stamps (100 MHz) per wavefront into Params::slow_list */
    unsigned* const tl_buf = (unsigned*)P.slow_list + ((blockIdx.x * 4u + wv) & 4095u) * 16u;
    int tl_it = 0;
    auto tl_stamp = [&](int k) { if (lane == 0u && k < 16) tl_buf[k] = (unsigned)__builtin_amdgcn_s_memrealtime(); };
    tl_stamp(0);
#endif
    QuadRaw nxt = issue(walk.first < walk.hi ? walk.first : -1);

    if (tid == 0u) S.next_quad = 4;
    if (DRAIN && tid == 0u) {
        S.local_count = 0;
        // the other control block still holds the PREVIOUS step's total (its launch is complete): report it to
        // the host (drain mode decision) and clear it for the next step
        if (blockIdx.x == 0u) queue_begin_drain(P, P.slow_count_next[0]);               // (a half launch of the pipelined mode has control blocks of its own)
    }
    // network tables, class multipliers and EVSE kinds: the chunk list the engine prepared (Params::cq_blob).  Addresses depend
    // on the thread id alone, so the list entry and its data are requested together: one round trip.
    {
        const CqBlob* const B = P.cq_blob;
        const unsigned count = B->count;
        for (unsigned i = tid; i < count; i += 256u) {
            const unsigned d = B->dst[i];
            const uint4 v = B->data[i];
            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(&S) + d) = v;
        }
    }
    {
        float4* const img4 = reinterpret_cast<float4*>(obs_img[wv][row]);
        img4[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        img4[16u + q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        reinterpret_cast<float4*>(act_img[wv][row])[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#pragma unroll
    for (int j = 0; j < kSlots; j++)
        if (DBG) dbg_img[wv][row][st4 + j] = 0.0;
    __syncthreads();

#ifdef EVC_TIMELINE
    tl_stamp(1);
#endif
    float* const obs_row = obs_img[wv][row];        // [demands n | est_departures n]
    float* const act_row = act_img[wv][row];
    double* const dbg_row = dbg_img[DBG ? wv : 0][row];

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
    // Cross-lane LDS hand-off inside the wave.  The LDS pipeline executes a wave's ds instructions in
    // issue order, so only the COMPILER must be kept from reordering them; a real fence would also
    // drain every outstanding global load and store (s_waitcnt vmcnt(0)) several times per step.
    auto lds_sync = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    };

    for (int quad = walk.first < walk.hi ? walk.first : -1; quad >= 0;) {
        int quad_next = -1;                                  // set where the next quad's loads are issued
        const unsigned env = (unsigned)quad * 4u + row;
        const bool ev = ALIVE || env < N;
        const unsigned ebase = env * n;
        const unsigned obase = env * F;                     // observation row
        const QuadRaw cur = nxt;
        EVC_TLP(0);
#if defined(EVC_TIMELINE) && EVC_TIMELINE < 3
        tl_stamp(2 + 2 * tl_it);
#if EVC_TIMELINE >= 2       /* how long until everything outstanding (prefetched rows, the previous quad's stores) is in */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tl_stamp(3 + 2 * tl_it);
#endif
#endif

        const v4u s0 = cur.s0, s1 = cur.s1;
        unsigned meta[kSlots];
        double rem[kSlots];
#pragma unroll
        for (int c = 0; c < kSlots; c++) { meta[c] = 0u; rem[c] = 0.0; }
        meta[0] = cur.meta0;
        rem[0] = cur.rem0;
        bool clamped = false;
        float a_st[kSlots];                         // clamped action of this lane's stations
#pragma unroll
        for (int j = 0; j < kSlots; j++) a_st[j] = 0.0f;
        {
            const float a_in[kSlots] = {__uint_as_float(cur.a4.x), __uint_as_float(cur.a4.y), __uint_as_float(cur.a4.z), __uint_as_float(cur.a4.w)};
#pragma unroll
            for (int j = 0; j < kSlots; j++) {
                const float a = a_in[j];
                a_st[j] = fminf(fmaxf(a, 0.0f), 1.0f);                    // NaN -> 0
                clamped = clamped | (st_valid[j] & (a_st[j] != a));       // also true for NaN (no short circuit: three exec-mask regions)
            }
            reinterpret_cast<float4*>(act_row)[q] = make_float4(a_st[0], a_st[1], a_st[2], a_st[3]);
        }
        // Without the projection the schedule of EMPTY stations is non-zero too and counts in the
        // constraint excess (env.py:449-452 evaluates the schedule, not the delivered rates): the
        // pilots' class sums are then taken on the station side.
        const bool station_pilots = !PROJECT && !greedy;
        // the accumulators are only needed with the reward: fetched in this iteration, not with the prefetch set
        // (two VGPRs fewer alive across iterations: 27.9 -> 27.5 us per step)
        const double acc_in = buf_ld_f64(r_acc, (ev && q < 3u) ? env * 24u + q * 8u : kOob);

        int t = (int)s0.x, cursor = (int)s0.y, slot = (int)s0.z, moer_day = (int)s0.w;
        int n_sessions = (int)s1.x, next_arrival = (int)s1.y, status = (int)s1.z & kStatusMask, episodes = (int)s1.w;
        const unsigned A = ev ? ((s1.z >> kCountShift) & 0x7fu) : 0u;
        const bool after_done = !ALIVE && ev && t >= EVC_EPISODE_STEPS;   // step() after termination w/o autoreset
        bool live = ev && !after_done;
        const int t1 = t + 1;
        const bool more = __ballot(A > 16u) != 0ull;            // wave-uniform: entry slots 1..3 in use
        // Round 4 — second-level loads requested as soon as the scalars are in (kEarly: the projecting kernels, which have the
        // registers at 3 wavefronts per SIMD; the non-projecting kernel at 4 per SIMD spills on them: 25.55 -> 26.3 us as one
        // launch, measured, so it keeps the late form).  (i) The session record at the cursor and the arrival time of the one
        // behind it: whether an EV arrives this period is known from the scalars alone (next_arrival), and the plug-in section
        // is a microsecond of work away — its two dependent round trips were exposed in the middle of the body (0.11 arrivals
        // per environment-step: one quad in three).  (ii) The MOER row of the period (reward, observation), formerly issued
        // after the charge section to spare four VGPRs under the 128-register cap.  Same-box A/B, three interleaved pairs:
        // (i) 23.17 -> 22.39 us per step pipelined, 26.79 -> 26.38 as one launch, GMM days 32.9 -> 32.05 / 45.5 -> 44.6;
        // (i)+(ii) 22.52 -> 22.29 / 26.32 -> 26.06.
        constexpr bool kEarly = PROJECT;
        double moer_now0 = 0.0;
        v4u mo0 = {0u, 0u, 0u, 0u};
        v2u sv0 = {0u, 0u};
        double rq0 = 0.0;
        unsigned nx0 = 0u;
        if constexpr (kEarly) {
            const unsigned mrow0 = live ? ((unsigned)moer_day * EVC_MOER_ROWS + (unsigned)t1) : 0u;
            moer_now0 = buf_ld_f64(r_hist, live ? mrow0 * 8u : kOob);
            mo0 = buf_ld_v4(r_mtail, (live && q < t_chunks) ? (mrow0 * mtail_w + st4) * 4u : kOob);   // chunk q of [forecast | prev | ts]
            const bool pend0 = live & (next_arrival <= t1) & (cursor < n_sessions);
            const unsigned sidx0 = (unsigned)slot * (unsigned)P.max_sessions + (unsigned)cursor;
            sv0 = buf_ld_v2(r_sess, pend0 ? sidx0 * 8u : kOob);
            rq0 = buf_ld_f64(r_req, pend0 ? sidx0 * 8u : kOob);
            nx0 = buf_ld_u32(r_sess, (pend0 & (cursor + 1 < n_sessions)) ? (sidx0 + 1u) * 8u : kOob);
        }
#ifdef EVC_PREFETCH_EARLY          /* measurement builds: the next quad's rows requested at the top of the iteration */
        quad_next = take_quad();
        if (quad_next >= 0) nxt = issue(quad_next);
#endif
        // The rest of the iteration is instantiated for NS = 1..4 entry slots per lane: the widest row of
        // the wave decides (NS = 1: every row has <= 16 entries, the normal case of a quiet network;
        // midday on a real Caltech / JPL day needs 2-3).  Registers — and spills, all of them in the
        // NS >= 2 copies — grow with NS, so a single general copy would make every busy hour pay for
        // 64 entries: 99-121 us per step at Caltech's midday with NS in {1,4}, 54-64 us with {1,2,3,4}.
        auto body = [&](auto ns_tag) {
        constexpr int NS = decltype(ns_tag)::value;
#pragma unroll
        for (int c = 1; c < NS; c++) {
            const unsigned e = (unsigned)c * 16u + q;
            meta[c] = buf_ld_u32(r_de, e < A ? (ebase + e) * 4u : kOob);
            rem[c] = buf_ld_f64(r_rem, e < A ? (ebase + e) * 8u : kOob);
        }

        EVC_TLP(1);
        // ---- entries: decode, action, y (box clip of the projection), class sums ----
        bool valid[kSlots];
        unsigned st[kSlots];
        int dep[kSlots], est[kSlots];
        float act[kSlots];
        double y[kSlots];
        unsigned ywords[WORDS], pwords[WORDS];
#pragma unroll
        for (int w = 0; w < WORDS; w++) { ywords[w] = 0u; pwords[w] = 0u; }
#pragma unroll
        for (int c = 0; c < kSlots; c++) { valid[c] = false; st[c] = 0u; dep[c] = kEmptyDep; est[c] = 0; act[c] = 0.0f; y[c] = 0.0; }
        lds_sync();                                 // action image written above
        auto decode = [&](int c) {
            valid[c] = (unsigned)c * 16u + q < A;
            st[c] = (unsigned)entry_station(meta[c]);
            dep[c] = valid[c] ? entry_dep(meta[c]) : kEmptyDep;
            est[c] = entry_est(meta[c]);
            rem[c] = valid[c] ? rem[c] : 0.0;
            float a = greedy ? ((rem[c] > Consts::FULLY_CHARGED_EPS) ? 1.0f : 0.0f) : act_row[st[c]];
            a = valid[c] ? a : 0.0f;
            act[c] = a;
            double yy = (double)a * Consts::ACTION_SCALE_FACTOR;        // env.py:366
            if (PROJECT) {
                yy = fmin(yy, quad_demand_cap(dep[c], rem[c]));
                const unsigned qy = (unsigned)(int)ceil(yy * 8.0);           // <= 256
                unsigned mw[WORDS];
                station_mulw(st[c], mw);
#pragma unroll
                for (int w = 0; w < WORDS; w++) ywords[w] = __umul24(qy, mw[w]) + ywords[w];
            }
            y[c] = yy;
        };
#pragma unroll
        for (int c = 0; c < NS; c++) decode(c);

        EVC_TLP(2);
        // ---- projection screen (PROJECT) ----
        bool pilots_screened = false;
        if (PROJECT) {
            row_allreduce_words<WORDS>(ywords);
            // Every lane evaluates a row — lanes q >= m repeat row 0, which changes no row_any() — so that the predicates handed
            // to the ballots are plain compares (LLVM lowers __ballot of anything else through v_cndmask + v_cmp) and no exec-mask
            // region surrounds the ten LDS reads.
            const float mag2 = quad_mag2_f32<WORDS>(net, qrow, ywords);
            const bool maybe = !(mag2 < net.thr_y2[qrow]);
            const bool maybe_p = !(mag2 < net.thr_yp2[qrow]);
            bool undecided = live & row_any(maybe, row);
            pilots_screened = !row_any(maybe_p, row);
#ifdef EVC_COUNT_UNDECIDED         /* diagnostic builds only: environments the screen leaves undecided, in metrics[7] */
            if (undecided && q == 0u) atomicAdd(P.tie_counters + 2 * (env & (kTieSlots - 1)) + 1, 1ull);
#endif
#ifdef EVC_ABL_NO_EXACT            /* ablation builds only (wrong results): cost of the exact path on congested days */
            undecided = false;
#endif
            if (__builtin_expect(__ballot(undecided) != 0ull, 0)) {
                // Rare (wave-uniform branch): exact float64 rows; class-cap (pod breaker) violations
                // are projected in closed form inside the row; anything else goes to the slow kernel.
                int st_gid[kSlots];
#pragma unroll
                for (int c = 0; c < kSlots; c++) st_gid[c] = valid[c] ? (int)(st_info[st[c]] & 0x7fu) : -1;
                unsigned cap_viol;
                bool hard = quad_exact_rows(P.G, P.class_cap, net, q, m, st_gid, y, undecided, cap_viol);
                bool anyviol = row_any(hard, row);
                // Class caps are filled whenever one is violated, also beside violated multi-class rows:
                // if the point projected onto box and caps satisfies every row it is the projection
                // (relaxation argument); what remains violated goes to the slow kernel.
                bool fill = undecided && cap_viol != 0u;
#ifdef EVC_ABL_NO_FILL              /* ablation builds only (wrong results): the exact rows without the water-filling */
                if (fill) anyviol = false;
                fill = false;
#endif
                if (__builtin_expect(__ballot(fill) != 0ull, 0)) {
                    bool slot_cc[kSlots];
#pragma unroll
                    for (int c = 0; c < kSlots; c++) slot_cc[c] = valid[c] && (st_info[st[c]] >> 7) != 0u;
                    for (int g = 0; g < P.G; g++) {
                        const bool do_g = fill && ((cap_viol >> g) & 1u);
#ifdef EVC_TRACE_FILL
                        if (__ballot(do_g) != 0ull) quad_waterfill(do_g, g, st_gid, act, dep, rem, P.class_cap[g], y, slot_cc, DBG ? P.tie_counters : nullptr, P.tie_log2, nullptr, env == (unsigned)(EVC_TRACE_FILL) && t == (EVC_TRACE_FILL_T));
#else
                        if (__ballot(do_g) != 0ull) quad_waterfill(do_g, g, st_gid, act, dep, rem, P.class_cap[g], y, slot_cc, DBG ? P.tie_counters : nullptr, P.tie_log2);
#endif
                    }
#ifdef EVC_ABL_NO_REVERIFY          /* ablation builds only (WRONG results): no second evaluation of the rows, a filled environment is never queued */
                    anyviol = anyviol && !fill;
#else
                    unsigned cv2;
                    const bool still = row_any(quad_exact_rows(P.G, P.class_cap, net, q, m, st_gid, y, fill, cv2, P.snap_tol), row);
                    anyviol = anyviol && !(fill && !still);
#endif
                }
                bool queue_me = undecided && anyviol;             // cones (or unsettled): slow kernel
                bool list_full = false;
                if (queue_me && q == 0u) {
                    if (DRAIN) {
                        // The list holds every environment the workgroup steps (the engine launches this form only with
                        // quads_per_wave * 16 <= kDrainListMax, also under EVC_DRAIN=1), so it cannot be full; should a future
                        // launch shape break that, the environment is stepped with what the row could settle and FLAGGED
                        // (EVC_STATUS_PROJ_NOCONV) instead of being silently skipped.
                        const int at = atomicAdd(&S.local_count, 1);
                        if (at < kDrainListMax) S.local_list[at] = (int)env;
                        else list_full = true;
                        __hip_atomic_fetch_add(P.slow_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // diagnostics only
                    } else {
                        queue_push(P, (int)env);
                    }
                }
                if (DRAIN) {
                    const bool full = row_any(list_full, row);
                    if (full) status |= EVC_STATUS_PROJ_NOCONV;
                    queue_me = queue_me && !full;
                }
                live = live && !queue_me;                 // queued rows write nothing here
                pilots_screened = pilots_screened && !undecided;
            }
        }

        EVC_TLP(3);
        // ---- pilots (env.py:366-378), battery charge ----
        double pilot[kSlots], amps[kSlots];
#pragma unroll
        for (int c = 0; c < kSlots; c++) { pilot[c] = 0.0; amps[c] = 0.0; }
        bool cross[kSlots];
        double rem_in[kSlots];
#pragma unroll
        for (int c = 0; c < kSlots; c++) { cross[c] = false; rem_in[c] = 0.0; }
        auto charge = [&](int c) {
            const unsigned info = st_info[st[c]];
            const double pl = legal_pilot(y[c], (info >> 7) != 0u);     // y = 0 on invalid entries
            pilot[c] = pl;
            rem_in[c] = rem[c];
            amps[c] = charge_ev_main(pl, rem[c], stepwise, cross[c]);    // every entry is a plugged-in EV
        };
#pragma unroll
        for (int c = 0; c < NS; c++) charge(c);
        // the period in which an EV reaches the ramp-down line (continuous battery model): once per session at most, so the
        // exponential sits behind a branch — the exec-mask region of the lanes concerned, which is skipped when there is none
        // (s_cbranch_execz; a wave-level __ballot in front of it cost two vector instructions per quad)
#pragma unroll
        for (int c = 0; c < NS; c++)
            if (__builtin_expect(cross[c], 0)) amps[c] = charge_ev_cross(pilot[c], rem_in[c], rem[c]);
        double amps_sum = 0.0;                     // env.py:445: delivered amps, summed over the entries (order: slot, then lane)
#pragma unroll
        for (int c = 0; c < NS; c++) amps_sum += (live && valid[c]) ? amps[c] : 0.0;
        double pilot_st[kSlots];                   // station-side pilots (station_pilots only)
#pragma unroll
        for (int j = 0; j < kSlots; j++) pilot_st[j] = 0.0;
        if (station_pilots) {
#pragma unroll
            for (int j = 0; j < kSlots; j++) {
                const unsigned s = (st4 + (unsigned)j) & 63u;
                const unsigned info = st_info[s];
                pilot_st[j] = st_valid[j] ? legal_pilot((double)a_st[j] * Consts::ACTION_SCALE_FACTOR, (info >> 7) != 0u) : 0.0;
                unsigned mw[WORDS];
                station_mulw(s, mw);
#pragma unroll
                for (int w = 0; w < WORDS; w++) pwords[w] = __umul24((unsigned)(int)pilot_st[j], mw[w]) + pwords[w];
            }
        }

        EVC_TLP(4);
        // ---- constraint excess of the pilots (env.py:449-452) ----
        // With the projection the y-screen above usually proves the pilots feasible as well
        // (pilots_screened); their class sums are only formed for wavefronts where some row is open.
        double excess = 0.0;
        if (station_pilots || (live && !pilots_screened)) {          // (row-uniform: whole 16-lane rows take part in the DPP sums)
            if (!station_pilots) {
#pragma unroll
                for (int c = 0; c < NS; c++) {
                    unsigned mw[WORDS];
                    station_mulw(st[c], mw);
#pragma unroll
                    for (int w = 0; w < WORDS; w++) pwords[w] = __umul24((unsigned)(int)pilot[c], mw[w]) + pwords[w];   // <= 32
                }
            }
            row_allreduce_words<WORDS>(pwords);
            bool maybe = false;
            if (q < m && !pilots_screened) maybe = !(quad_mag2_f32<WORDS>(net, q, pwords) < net.thr_p2[q]);
            if (__builtin_expect(__ballot(maybe && live) != 0ull, 0)) {   // rare: exact evaluation
                double ex = 0.0;
                if (q < m) ex = fmax(row_mag_f64<WORDS>(net, (int)q, pwords, 1.0) - net.mag[q], 0.0);
                excess = row_allreduce_f64(ex);
            }
        }

        EVC_TLP(5);
        // ---- acnsim event pass at iteration t1: unplug (precedence 0) before plug-in (10) ----
        // survivors keep their relative order; pos = index of the entry in the packed list
        bool alive[kSlots];
        unsigned pos[kSlots];
        unsigned count = 0u;                                             // row-uniform
#pragma unroll
        for (int c = 0; c < kSlots; c++) { alive[c] = false; pos[c] = 0u; }
        auto pack = [&](int c) {
            alive[c] = dep[c] > t1;                                      // (dep is kEmptyDep = -1 where the slot holds no entry: a plain compare)
            const unsigned bits = (unsigned)(__ballot(alive[c]) >> (row * 16u)) & 0xffffu;
            pos[c] = count + (unsigned)__popc(bits & ((1u << q) - 1u));
            count += (unsigned)__popc(bits);
        };
#pragma unroll
        for (int c = 0; c < NS; c++) pack(c);

        // MOER loads for t1 (row-uniform addresses); issued here rather than at the top of the iteration: the values are
        // only stored at its end, and four VGPRs fewer alive through the charge section are worth 0.8 us per step
        double moer_now = moer_now0;
        v4u mo = mo0;
        if constexpr (!kEarly) {
        const unsigned mrow = live ? ((unsigned)moer_day * EVC_MOER_ROWS + (unsigned)t1) : 0u;
        moer_now = buf_ld_f64(r_hist, live ? mrow * 8u : kOob);
        mo = buf_ld_v4(r_mtail, (live && q < t_chunks) ? (mrow * mtail_w + st4) * 4u : kOob);       // chunk q of [forecast | prev | ts]
        }

        if (live) t = t1;
        unsigned long long arrived = 0ull;                               // stations plugged in this pass
        bool pending = live && next_arrival <= t1 && cursor < n_sessions;
        bool first_pass = kEarly;                   // the first record of the period was requested at the top of the iteration
        while (pending) {                                                // (row-uniform; rows without an arrival sit the loop out)
            const unsigned sidx = (unsigned)slot * (unsigned)P.max_sessions + (unsigned)cursor;
            v2u sv;
            double rq;
            if (first_pass) { sv = sv0; rq = rq0; }            // (wave-uniform)
            else { sv = buf_ld_v2(r_sess, pending ? sidx * 8u : kOob); rq = buf_ld_f64(r_req, pending ? sidx * 8u : kOob); }
            const int s_dep = (int)(short)(sv.x >> 16);
            const int s_est = (int)(short)(sv.y & 0xffffu);
            const unsigned s_st = (sv.y >> 16) & 63u;
            bool busy = false;
#pragma unroll
            for (int c = 0; c < kSlots; c++) busy = busy || (pending && alive[c] && st[c] == s_st);
            const bool row_busy = row_any(busy, row) || ((arrived >> s_st) & 1ull);
            if (pending && row_busy) status |= EVC_STATUS_OCCUPIED;      // acnportal: StationOccupiedError
            const bool plug = pending && !row_busy;
            if (plug) {
                if (q == 0u) {
                    buf_st_u32(r_de, (ebase + count) * 4u, pack_entry(s_dep, (int)s_st, s_est));
                    buf_st_f64(r_rem, (ebase + count) * 8u, rq);
                    const bool active = rq > Consts::FULLY_CHARGED_EPS;
                    obs_row[s_st] = active ? (float)rq : 0.0f;
                    obs_row[n + s_st] = active ? (float)(s_est - t) : 0.0f;
                }
                count += 1u;
                arrived |= 1ull << s_st;
            }
            if (pending) {
                cursor += 1;
                next_arrival = kNoArrival;
            }
            const bool more_ev = pending && cursor < n_sessions;
            const unsigned nx = first_pass ? nx0 : buf_ld_u32(r_sess, more_ev ? (sidx + 1u) * 8u : kOob);
            first_pass = false;
            if (more_ev) next_arrival = (int)(short)(nx & 0xffffu);
            pending = more_ev && next_arrival <= t1;
        }
        if (live && row_any(clamped, row)) status |= EVC_STATUS_ACTION_CLAMPED;
        const bool done = live && t1 >= EVC_EPISODE_STEPS;
        if (after_done) status |= EVC_STATUS_STEP_AFTER_DONE;

        if (DBG) {                                    // per-station debug / parity outputs, dense [N][n]
            double* outs[3] = {io.out.pilots, io.out.rates, io.out.projected};
#pragma unroll
            for (int o = 0; o < 3; o++) {
                if (!outs[o]) continue;
                if (station_pilots && o != 1) {               // schedule of every station, plugged or not
#pragma unroll
                    for (int j = 0; j < kSlots; j++)
                        if (live && st_valid[j])
                            outs[o][(size_t)ebase + st4 + (unsigned)j] = o == 0 ? pilot_st[j] : (double)a_st[j];
                    continue;
                }
#pragma unroll
                for (int c = 0; c < kSlots; c++)
                    if (live && valid[c])
                        dbg_row[st[c]] = o == 0 ? pilot[c] : (o == 1 ? amps[c] : y[c] / Consts::ACTION_SCALE_FACTOR);
                lds_sync();
#pragma unroll
                for (int j = 0; j < kSlots; j++) {
                    const double v = dbg_row[(st4 + (unsigned)j) & 63u];
                    if (live && st_valid[j]) outs[o][(size_t)ebase + st4 + (unsigned)j] = v;
                }
                lds_sync();
#pragma unroll
                for (int c = 0; c < kSlots; c++)
                    if (live && valid[c]) dbg_row[st[c]] = 0.0;
                lds_sync();
            }
        }

        // prefetch of the next quad's rows: issued here, after the charge / event section (its 15 VGPRs are not alive through
        // the register-hungry part of the iteration: 184 -> 155 spilled VGPRs, -0.7 us per step with synchronised phases);
        // nothing to fetch after the last quad
#if defined(EVC_TIMELINE) && EVC_TIMELINE < 2
        tl_stamp(3 + 2 * tl_it);
#endif
        EVC_TLP(6);
#ifndef EVC_PREFETCH_EARLY
        quad_next = take_quad();
#ifdef EVC_ISSUE_ALWAYS      /* measurement builds: behind the last quad five out-of-range loads instead of a merge with the old rows (level) */
        nxt = issue(quad_next);
#else
        if (quad_next >= 0) nxt = issue(quad_next);
#endif
#endif
        EVC_TLP(7);
        // ---- observation image: demands / est_departures of the surviving entries ----
        auto scatter_obs = [&](int c) {
            if (live && alive[c]) {
                const bool active = rem[c] > Consts::FULLY_CHARGED_EPS;
                obs_row[st[c]] = active ? (float)rem[c] : 0.0f;
                obs_row[n + st[c]] = active ? (float)(est[c] - t) : 0.0f;
            }
        };
#pragma unroll
        for (int c = 0; c < NS; c++) scatter_obs(c);
        lds_sync();
        float4 d[2];                                // chunks q and q + 16 of [demands | est_departures]
#pragma unroll
        for (int p = 0; p < 2; p++) {
            float4* const chunk = reinterpret_cast<float4*>(obs_row) + ((unsigned)p * 16u + q);
            d[p] = *chunk;
            *chunk = make_float4(0.0f, 0.0f, 0.0f, 0.0f);       // leave the image clean for the next step
        }
        const double total_rate = row_allreduce_f64(amps_sum);           // env.py:445

        EVC_TLP(8);
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

        // The observation row leaves as 16-byte chunks: [demands | est_departures] from the image (d[0], d[1]), the tail
        // [forecasted_moer | prev_moer | timestep] from the table row (mo).  Whole chunks are dwordx4 stores; what the row
        // lengths leave over (2 floats of the first part when n is odd, (k + 2) mod 4 of the tail) goes behind wave-uniform
        // branches: for n = 54, k = 36 that is 2 + 1 stores of 16 bytes and one of 8.
        auto store_obs = [&](const rsrc_t& r, bool w) {
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const unsigned c = (unsigned)p * 16u + q;
                const v4u v = {__float_as_uint(d[p].x), __float_as_uint(d[p].y), __float_as_uint(d[p].z), __float_as_uint(d[p].w)};
                buf_st_v4(r, (w && c < c_full) ? (obase + c * 4u) * 4u : kOob, v);
                if (n & 1u) buf_st_v2(r, (w && c == c_full) ? (obase + c * 4u) * 4u : kOob, v2u{v.x, v.y});
            }
            const unsigned tb = (obase + 2u * n + st4) * 4u;
            buf_st_v4(r, (w && q < t_full) ? tb : kOob, mo);
            if (t_rem & 2u) buf_st_v2(r, (w && q == t_full) ? tb : kOob, v2u{mo.x, mo.y});
            if (t_rem & 1u) buf_st_u32(r, (w && q == t_full) ? tb + (t_rem & 2u) * 4u : kOob, (t_rem & 2u) ? mo.z : mo.x);
        };

        EVC_TLP(9);
        // ---- autoreset (gymnasium VectorEnv): terminal observation, then next episode's state ----
        const bool do_reset = done && P.autoreset;
        if (done) episodes += 1;
        if (__builtin_expect(do_reset, 0)) {                             // (row-uniform; per-lane work only)
            if (io.out.final_obs) {
                const rsrc_t r_fin = row_rsrc(io.out.final_obs, N * F * 4u);
                store_obs(r_fin, do_reset);
            }
            if (do_reset) {
                const int next = (slot + P.autoreset_stride) % P.bank_slots;
                slot = next;
                t = 0; cursor = 0;
                // three independent loads, issued together (one round trip; a load behind `n_sessions > 0` made it two,
                // and with staggered episode phases some wavefront pays this tail in every launch)
                const int first_arrival = (int)P.sessions[(size_t)next * P.max_sessions].arrival;
                moer_day = P.slot_moer_day[next];
                n_sessions = P.n_sessions[next];
                // (the waits for these three stay in this block, like the one for the table row below)
                asm volatile("" ::"v"(first_arrival), "v"(moer_day), "v"(n_sessions));
                next_arrival = n_sessions > 0 ? first_arrival : kNoArrival;
                count = 0u;
                d[0] = d[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            // first observation of the next episode: the table's row of period 0 (timestep 0)
            const unsigned mrow0 = (unsigned)moer_day * EVC_MOER_ROWS;
            const v4u v = buf_ld_v4(r_mtail, (do_reset && q < t_chunks) ? (mrow0 * mtail_w + st4) * 4u : kOob);
            // The wait for this load belongs in here.  vmcnt retires in order and a wait is computed for the shortest history any path
            // has behind the load: left to the first use of `mo` (the observation's stores, behind the join) this load's wait became that
            // use's wait on EVERY path — vmcnt(1) in the middle of the common path's stores, i.e. every iteration stood until the stores
            // it had just issued were acknowledged (tools/isa_walk.py; -1.2 % step period, -2 % as one launch, profiles/r6_ab_vmcnt*.txt)
            asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            if (do_reset) mo = v;
        }

        EVC_TLP(10);
        // ---- observation (env.py:381-394) + state write-back ----
        store_obs(r_obs, live);
        auto store_entry = [&](int c) {
            const bool w = live && alive[c] && !do_reset;
            buf_st_u32(r_de, w ? (ebase + pos[c]) * 4u : kOob, meta[c]);
            buf_st_f64(r_rem, w ? (ebase + pos[c]) * 8u : kOob, rem[c]);
        };
#pragma unroll
        for (int c = 0; c < NS; c++) store_entry(c);
        buf_st_f64(r_acc, (live && q < 3u) ? env * 24u + q * 8u : kOob, do_reset ? 0.0 : acc);
        {
            v4u o0, o1;
            o0.x = (unsigned)t; o0.y = (unsigned)cursor; o0.z = (unsigned)slot; o0.w = (unsigned)moer_day;
            o1.x = (unsigned)n_sessions; o1.y = (unsigned)next_arrival;
            o1.z = ((unsigned)status & (unsigned)kStatusMask) | ((live ? count : A) << kCountShift);
            o1.w = (unsigned)episodes;
            const unsigned so = ((live || after_done) && q == 0u) ? env * 32u : kOob;
            buf_st_v4(r_scal, so, o0);
            buf_st_v4(r_scal, so == kOob ? kOob : so + 16u, o1);
        }
        };
#ifdef EVC_ABL_NS1_ONLY      /* ablation builds only (WRONG results beyond 16 EVs per environment): what do the wider copies cost the NS = 1 path? */
        if (false) {
#else
        if (__builtin_expect(more, 0)) {
#endif
            if (__ballot(A > 48u) != 0ull) body(std::integral_constant<int, 4>{});
            else if (__ballot(A > 32u) != 0ull) body(std::integral_constant<int, 3>{});
            else body(std::integral_constant<int, 2>{});
        } else {
            body(std::integral_constant<int, 1>{});
        }
        EVC_TLP(11);
        lds_sync();
        quad = quad_next;
#ifdef EVC_TIMELINE
        tl_it++;
#endif
    }
#ifdef EVC_TIMELINE
    tl_stamp(14);
#endif

    if (DRAIN) {
        __syncthreads();                       // every wave's queue entries are in the list; the images are free
        const int count = S.local_count < kDrainListMax ? S.local_count : kDrainListMax;
        if (__builtin_expect(count != 0, 0) && wv == 0u) {
#ifndef EVC_ABL_DRAIN_NO_SOLVE      /* ablation builds only: the tail's own cost without the slow path's code */
            drain_local_list<WORDS>((unsigned)(size_t)(__attribute__((address_space(3))) void*)&S);
#endif
        }
    }
#ifdef EVC_TIMELINE
    tl_stamp(15);
#endif
}

}  // namespace evc
