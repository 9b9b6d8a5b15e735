// evc_engine.hip — C-ABI implementation (include/evcharge.h) of the MI355X-native batched
// EV-charging step engine.  Host-side bookkeeping only; all simulation arithmetic is in the
// gfx950 kernels of evc_cquad.h (default streaming kernel), evc_quad.h, evc_kernels.h, evc_solver.h and
// evc_gen.h (episode generator).  There is no CPU execution path.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

#include "evc_hostcopy.h"

#include "evc_solver.h"
#include "evc_quad.h"
#include "evc_cquad.h"
#include "evc_rollout_launch.h"
#include "evc_gen.h"

using namespace evc;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return fail(EVC_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                              \
    } while (0)

template <typename T>
hipError_t dmalloc(T** p, size_t count) {
    return hipMalloc((void**)p, sizeof(T) * (count ? count : 1));
}

}  // namespace

constexpr bool kDefaultCompact = true;
constexpr int kQlenRing = 288;           // one report per period of a day
#ifndef EVC_PROJ_WAVES
#define EVC_PROJ_WAVES 3                // wavefronts per SIMD the PROJECTING lean compact kernels are held to (evc_cquad.h, WAVES)
#endif
constexpr int kDrainLookBack = 12, kDrainLookAhead = 36;   // periods of the day around the current one whose reports decide the step's mode
constexpr int kPipeSkewUs = 0;             // start skew (us) of the second half of a pipelined step train (launch_split): measured twice, no reliable gain
constexpr int kDrainMaxQueuePipelined = 256; // ... pipelined halves: a workgroup's drain runs under the other half's launches (JPL GMM days 48.4 -> 46.9 us per step)
constexpr int kDrainMaxQueueDefault = 16;   // in-kernel drain only while NO step of the last day queued more than this (EVC_DRAIN_MAXQ overrides)

struct evc_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    Params P{};
    uint32_t flags = 0;
    // owned device buffers
    char* d_arena = nullptr;      // one block behind the pointers down to d_tables (struct Win window)
    double* d_rem = nullptr;
    int* d_depest = nullptr;
    int4* d_scal = nullptr;
    double* d_acc = nullptr;
    evc_session* d_sessions = nullptr;
    double* d_requested = nullptr;
    int* d_nsess = nullptr;
    int* d_slot_moer = nullptr;
    double* d_moer_hist = nullptr;
    float* d_moer_obs = nullptr;
    unsigned* d_roll_order = nullptr;     // quads of a fused rollout launch, busiest first (RolloutIO::quad_order)
    unsigned* d_roll_ctr = nullptr;       // next unassigned quad of a fused rollout launch (RolloutIO::quad_counter)
    CqBlob* d_cq_blob = nullptr;         // prologue image of the compact streaming kernel (Params::cq_blob)
    float* d_moer_tail = nullptr;        // [moer_days][289][mtail_w]: the observation row's tail, ready to store (Params::off_mtail)
    NetTables* d_tables = nullptr;
    int* d_slow_count = nullptr;  // [2 halves][2]: queue length per step parity (second pair: the second half launch of the pipelined mode); [4..5]: always zero, for the warm-up launches
    int* d_slow_list = nullptr;
    unsigned long long* d_tie = nullptr;   // Params::tie_counters
    char* h_act = nullptr;                 // page-locked staging block for pageable host actions (evc_step_host, direct mode)
    // Drain mode (who solves what the streaming kernel queues): on a workload whose steps queue at most a few
    // dozen environments every workgroup of the lean compact streaming kernel solves the ones it queued itself
    // and NO slow kernel is launched (saves the ~2 us an almost always empty dependent launch costs per step;
    // solves of different workgroups run side by side).  Where steps queue more — several per workgroup would
    // serialise — the slow kernel runs (one workgroup per queued environment).  The decision reads the per-step
    // totals the kernels report into a page-locked ring of one day: stale by however far the host runs ahead,
    // which only costs speed, never correctness — either drainer finishes every queued step.
    int* h_qlen = nullptr;        // [2 halves][kQlenRing] host view (a step that is one launch reports in the first ring, the host zeroes its slot of the second)
    int* d_qlen = nullptr;        // device address of the same memory
    unsigned long long step_index = 0;
    bool warmed = false;          // both lean streaming copies have been launched once
    int drain_override = -1;      // EVC_DRAIN=0/1 forces a mode (measurements)
    int drain_max_queue = kDrainMaxQueueDefault;
    int drain_max_queue_pipelined = kDrainMaxQueuePipelined;
    int* d_idbuf = nullptr;       // reset ids/slots staging [2N]
    double* d_metrics = nullptr;  // [8]
    double* d_maxprofit = nullptr;  // [bank_slots] env.py:422-429 of the episode in each slot
    GenTables* d_gen = nullptr;     // episode-generator model (evc_upload_gmm)
    // device staging for the *_host entry points
    float* d_act_f32 = nullptr;   // float32 actions produced by discretize_kernel
    void* d_act = nullptr;        // N*n*8 bytes
    float* d_obs = nullptr;
    double* d_reward = nullptr;
    uint8_t* d_term = nullptr;
    double* d_breakdown = nullptr;
    float* d_final = nullptr;
    double* d_pilots = nullptr;
    double* d_rates = nullptr;
    double* d_proj = nullptr;
    // timing
    bool timing = false;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // main start/stop, slow start/stop, second half start/stop
    bool ev_valid = false, ev_slow = false, ev_split = false;
    // Pipelined halves (evc_set_pipeline(e, 2)): the lean compact streaming kernel steps the batch as two launches —
    // quads [0, mid) and [mid, end) — on two side streams.  Both wait for what the engine's stream holds at the call (the
    // caller's actions; nothing to wait for if that stream is idle), consecutive steps of one half are ordered by their
    // stream, the two halves are not ordered with each other: the tail of one launch (a few wavefronts in the rare
    // projection branch) runs under the body of the other half's launches.  The engine's stream waits
    // for the side streams at the next join (evc_join, or any other engine call — bind() joins).
    int pipeline = 1;
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t fork_ev = nullptr, join_ev[2] = {nullptr, nullptr};
    bool halves_pending = false, side_warmed = false, last_split = false, side_ready = false;
    bool halves_exposed = false;  // evc_pipeline_half handed the side streams to the caller (closed loop per half)
    unsigned train_len = 0, prev_train_len = 0;   // pipelined steps since the last join, and in the train before it
    std::chrono::steady_clock::time_point last_split_issue{};   // host time of the last pipelined step's issue
    unsigned long long split_steps = 0;   // steps that ran as two half launches (evc_pipelined_steps) ...
    unsigned long long fork_steps = 0;    // ... and how many of them had to be ordered behind pending work of the engine's stream
    // fused rollout: which register budget of the projecting kernels is faster on the caller's workload (launch_rollout)
    struct RolloutTuner {
        hipEvent_t ev[2] = {nullptr, nullptr};
        int pending = -1;                 // setting (2 | 3) of the launch whose events are outstanding
        double pending_work = 0.0;        // its environment-steps
        double best[2] = {0.0, 0.0};      // ms per environment-step: [0] two, [1] three wavefronts per SIMD
        int count[2] = {0, 0};
        unsigned launches = 0;
    } roll;
    int last_rollout_waves = 0;
    // host mirrors
    unsigned long long env_steps = 0;
    int step_parity = 0;
    int num_cus = 256;
    int step_grid = 0, solver_grid = 0, quad_grid = 0, proj_grid = 0;
    unsigned long long policy_seed = 0;   // EVC_ACTION_RANDOM (evc_set_policy_seed)
    unsigned env_id_base = 0;
    bool compact = false;        // state layout (Params::compact)
    int side_streams_replaced = 0;      // evc_set_pipeline(2): second-half streams rejected because they did not run beside the first half's
    bool side_streams_overlap = true;   // ... and whether the pair in use does
    bool site_kernels = true;    // lean kernels with the site's shape compiled in where it matches (EVC_SITE_KERNELS=0 at evc_create: never)
    bool clocks_in_range = true; // every environment's t is in [0, 288): true for the zeroed state, kept by reset and by stepping under
                                 // autoreset; evc_set_env_scalars re-evaluates it (step_kernel_cquad's ALIVE)
    bool use_quad = false;
};

namespace {

__global__ void side_stream_warmup_kernel() {}

// ... and reports when it began and ended (evc_set_pipeline: do the two side streams run concurrently?)
__global__ void spin_stamp_kernel(unsigned long long* out, unsigned ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[0] = t0; out[1] = wall_clock64(); }
}

// one wavefront that does nothing for `ticks` of the 100 MHz constant clock (launch_split: start skew of the second half)
__global__ void skew_kernel(unsigned ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// the engine's stream waits for the two half launches of the pipelined mode (no-op otherwise)
int join_halves(evc_engine* e) {
    if (!e->halves_pending) return EVC_OK;
    for (int h = 0; h < 2; h++) {
        HIP_TRY(hipEventRecord(e->join_ev[h], e->side[h]));
        HIP_TRY(hipStreamWaitEvent(e->stream, e->join_ev[h], 0));
    }
    e->halves_pending = false;
    return EVC_OK;
}

int bind_device(evc_engine* e) {
    HIP_TRY(hipSetDevice(e->device));
    return EVC_OK;
}

// every entry point but evc_step: whatever it enqueues or reads is ordered after the pending half launches
int bind(evc_engine* e) {
    HIP_TRY(hipSetDevice(e->device));
    return join_halves(e);
}

void free_all(evc_engine* e) {
    void* ptrs[] = {e->d_arena /* rem, depest, scal, acc, sessions, requested, moer tables, net tables */,
                    e->d_nsess, e->d_slot_moer,
                    e->d_slow_count, e->d_slow_list, e->d_idbuf, e->d_metrics, e->d_act, e->d_act_f32, e->d_obs,
                    e->d_reward, e->d_term, e->d_breakdown, e->d_final, e->d_pilots, e->d_rates,
                    e->d_proj, e->d_maxprofit, e->d_gen, e->d_tie, e->d_cq_blob, e->d_roll_ctr, e->d_roll_order};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (auto& ev : e->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : e->roll.ev)
        if (ev) (void)hipEventDestroy(ev);
    if (e->fork_ev) (void)hipEventDestroy(e->fork_ev);
    for (int h = 0; h < 2; h++) {
        if (e->join_ev[h]) (void)hipEventDestroy(e->join_ev[h]);
        if (e->side[h]) (void)hipStreamDestroy(e->side[h]);
    }
    if (e->h_qlen) (void)hipHostFree(e->h_qlen);
    if (e->h_act) (void)hipHostFree(e->h_act);
}

// Station classes: identical (constraint column, phase angle).  Every constraint depends on a
// schedule only through the per-class sums (what the kernels reduce over).
int build_tables(const evc_network_desc* net, Params& P, NetTables& T) {
    const int n = net->n_stations, m = net->n_constraints;
    std::vector<int> rep;
    std::vector<int> gid(n, -1);
    for (int i = 0; i < n; i++) {
        int found = -1;
        for (size_t g = 0; g < rep.size() && found < 0; g++) {
            const int j = rep[g];
            bool same = net->phase_angles_deg[i] == net->phase_angles_deg[j];
            for (int c = 0; c < m && same; c++)
                same = net->constraint_matrix[c * n + i] == net->constraint_matrix[c * n + j];
            if (same) found = (int)g;
        }
        if (found < 0) {
            if ((int)rep.size() >= EVC_MAX_GROUPS)
                return fail(EVC_EINVAL, "network has more than %d station classes", EVC_MAX_GROUPS);
            found = (int)rep.size();
            rep.push_back(i);
        }
        gid[i] = found;
    }
    P.G = (int)rep.size();
    memset(&T, 0, sizeof(T));
    for (int g = 0; g < EVC_MAX_GROUPS; g++) P.group_mask[g] = 0ull;
    P.cc_mask = 0ull;
    for (int i = 0; i < n; i++) {
        P.group_mask[gid[i]] |= 1ull << i;
        if (net->evse_kind[i] == EVC_EVSE_CC) P.cc_mask |= 1ull << i;
        else if (net->evse_kind[i] != EVC_EVSE_AV)
            return fail(EVC_EINVAL, "evse_kind[%d] = %d is not EVC_EVSE_AV/CC", i, net->evse_kind[i]);
    }
    for (int t = 0; t < EVC_MOER_ROWS; t++)
        T.timestep[t] = (float)((double)t / (double)EVC_EPISODE_STEPS);         // env.py:392
    std::vector<double> slack(m, 0.0), round_slack(m, 0.0);
    for (int g = 0; g < P.G; g++) {
        const int j = rep[g];
        const double rad = net->phase_angles_deg[j] * (M_PI / 180.0);  // np.deg2rad, env.py:485
        const double cs = std::cos(rad), sn = std::sin(rad);
        const int n_g = __builtin_popcountll(P.group_mask[g]);
        for (int c = 0; c < m; c++) {
            const double a = net->constraint_matrix[c * n + j];
            T.Mre[g][c] = a * cs;
            T.Mim[g][c] = a * sn;
            T.Mre32[g][c] = (float)(a * cs);
            T.Mim32[g][c] = (float)(a * sn);
            slack[c] += std::fabs(a) * n_g / 8.0;   // quantisation of ceil(8 y) per station
            // env.py:373-378 rounding can raise a station by 0.5 A (AV) / 4 A (CC)
            const int n_cc = __builtin_popcountll(P.group_mask[g] & P.cc_mask);
            round_slack[c] += std::fabs(a) * (0.5 * (n_g - n_cc) + 4.0 * n_cc);
        }
    }
    for (int c = 0; c < m; c++) {
        if (!(net->magnitudes[c] > 0.0)) return fail(EVC_EINVAL, "magnitudes[%d] must be > 0", c);
        const double r = net->magnitudes[c];
        T.mag[c] = r;
        const double ry = (r - slack[c]) * (1.0 - 1e-4) * 8.0;
        T.thr_y2[c] = ry > 0.0 ? (float)(ry * ry) : 0.0f;
        const double rp = r * (1.0 - 1e-4);
        T.thr_p2[c] = (float)(rp * rp);
        const double ryp = (r - slack[c] - round_slack[c]) * (1.0 - 1e-4) * 8.0;
        T.thr_yp2[c] = ryp > 0.0 ? (float)(ryp * ryp) : 0.0f;
    }
    // step of the slow path's proximal-gradient safeguard (evc_solver.h): B = rows [Re; Im] of A~ in station
    // space, (B B')_ab = sum_g n_g M_a[g] M_b[g]; lambda_max <= largest absolute row sum (Gershgorin)
    {
        double bound = 0.0;
        for (int a = 0; a < 2 * m; a++) {
            double row = 0.0;
            for (int b = 0; b < 2 * m; b++) {
                double dot = 0.0;
                for (int g = 0; g < P.G; g++) {
                    const double ma = (a & 1) ? T.Mim[g][a >> 1] : T.Mre[g][a >> 1];
                    const double mb = (b & 1) ? T.Mim[g][b >> 1] : T.Mre[g][b >> 1];
                    dot += (double)__builtin_popcountll(P.group_mask[g]) * ma * mb;
                }
                row += std::fabs(dot);
            }
            bound = std::max(bound, row);
        }
        P.prox_step = bound > 0.0 ? 1.0 / bound : 0.0;
    }
    // Rows are re-verified after the tie snap (2^-16 A grid: a value moves by at most 2^-17 A): the snap can add
    // sum_g |A_cg| n_g 2^-17 A to row c and n_g 2^-17 A to the sum of class g
    {
        double worst = 0.0;
        for (int c = 0; c < m; c++) {
            double load = 0.0;
            for (int g = 0; g < P.G; g++)
                load += std::fabs(net->constraint_matrix[c * n + rep[g]]) * __builtin_popcountll(P.group_mask[g]);
            worst = std::max(worst, load / net->magnitudes[c]);
        }
        P.snap_load = worst;                                   // amps-of-row per amp-of-station over the limit, worst row
        P.tie_log2 = Consts::TIE_LOG2;
        P.snap_tol = Consts::PROJ_TOL + worst * std::ldexp(1.0, -(P.tie_log2 + 1));
    }
    // simple rows: all non-zero coefficients inside one station class -> a cap on that class sum
    P.simple_rows = 0u;
    for (int g = 0; g < EVC_MAX_GROUPS; g++) P.class_cap[g] = HUGE_VAL;
    for (int c = 0; c < m; c++) {
        int cls = -1;
        bool simple = true;
        for (int i = 0; i < n && simple; i++) {
            if (net->constraint_matrix[c * n + i] == 0.0) continue;
            if (cls < 0) cls = gid[i];
            else if (cls != gid[i]) simple = false;
        }
        if (simple && cls >= 0) {
            // a class has identical columns, so every member carries the same coefficient
            const double a = std::fabs(net->constraint_matrix[c * n + rep[cls]]);
            // the row must load the WHOLE class (true by construction of the classes)
            P.simple_rows |= 1u << c;
            const double cap = net->magnitudes[c] / a;
            if (cap < P.class_cap[cls]) P.class_cap[cls] = cap;
        }
    }
    P.cap_classes = 0u;
    for (int g = 0; g < P.G; g++)
        if (P.class_cap[g] < HUGE_VAL) P.cap_classes |= 1u << g;
    // monotone rows: |sum_g M_c[g] S_g|^2 = sum_gh Re(M_c[g] conj M_c[h]) S_g S_h does not decrease in any S_g >= 0 if no
    // pair of the row's class phasors has a negative inner product
    P.monotone_rows = 1;
    for (int c = 0; c < m; c++)
        for (int g = 0; g < P.G; g++)
            for (int h = g + 1; h < P.G; h++)
                if (T.Mre[g][c] * T.Mre[h][c] + T.Mim[g][c] * T.Mim[h][c] < -1e-12) P.monotone_rows = 0;
    if (const char* s = getenv("EVC_CAPS_SHORTCUT")) P.monotone_rows = P.monotone_rows && atoi(s) != 0;   // measurements: 0 disables the shortcut
    return EVC_OK;
}

// Prologue image of step_kernel_cquad (Params::cq_blob): the [Gp][m] corner of the network tables and the per-station
// class multipliers / EVSE kinds exactly as the kernel's prologue used to derive them, cut into 16-byte chunks with the
// byte offset each one has inside the workgroup's CquadLds.
void build_cq_blob(const Params& P, const NetTables& T, CqBlob& B) {
    memset(&B, 0, sizeof(B));
    unsigned cnt = 0;
    auto put_range = [&](size_t dst, const void* src, size_t bytes) {          // whole chunks; rows are wide enough to be over-read
        for (size_t o = 0; o < bytes; o += 16) {
            memcpy(&B.data[cnt], (const char*)src + o, 16);
            B.dst[cnt++] = (unsigned)(dst + o);
        }
    };
    const size_t m = (size_t)P.m, net0 = offsetof(CquadLds, net);
    const int Gp = (P.G + 1) & ~1;                  // classes are padded to an even count (two per packed word); the pad row is zero
    for (int g = 0; g < Gp; g++) {
        put_range(net0 + offsetof(LdsNet, Mre) + sizeof(T.Mre[0]) * g, T.Mre[g], m * 8);
        put_range(net0 + offsetof(LdsNet, Mim) + sizeof(T.Mim[0]) * g, T.Mim[g], m * 8);
        put_range(net0 + offsetof(LdsNet, Mre32) + sizeof(T.Mre32[0]) * g, T.Mre32[g], m * 4);
        put_range(net0 + offsetof(LdsNet, Mim32) + sizeof(T.Mim32[0]) * g, T.Mim32[g], m * 4);
    }
    put_range(net0 + offsetof(LdsNet, mag), T.mag, m * 8);
    put_range(net0 + offsetof(LdsNet, thr_y2), T.thr_y2, m * 4);
    put_range(net0 + offsetof(LdsNet, thr_p2), T.thr_p2, m * 4);
    put_range(net0 + offsetof(LdsNet, thr_yp2), T.thr_yp2, m * 4);
    uint4 lo[64], hi[64];
    unsigned char info[64];
    for (unsigned s = 0; s < 64; s++) {
        const bool valid = s < (unsigned)P.n;
        int gid = 0;
        for (int g = 0; g < P.G; g++)
            if ((P.group_mask[g] >> s) & 1ull) gid = g;
        unsigned mw[8];
        for (int w = 0; w < 8; w++) mw[w] = (valid && (gid >> 1) == w) ? ((gid & 1) ? 65536u : 1u) : 0u;
        lo[s] = make_uint4(mw[0], mw[1], mw[2], mw[3]);
        hi[s] = make_uint4(mw[4], mw[5], mw[6], mw[7]);
        info[s] = (unsigned char)((unsigned)gid | ((unsigned)((P.cc_mask >> s) & 1ull) << 7));
    }
    put_range(offsetof(CquadLds, st_mulw), lo, sizeof(lo));
    if (P.G > 8) put_range(offsetof(CquadLds, st_mulw_hi), hi, sizeof(hi));      // packed words 4..7 exist only beyond eight classes
    put_range(offsetof(CquadLds, st_info), info, sizeof(info));
    B.count = cnt;
}

// Params describes exactly the shape the site kernels have compiled in (step_kernel_cquad's / rollout_kernel's NC)
bool site_shape_ok(const evc_engine* e, int nc) {
    return nc != 0 && e->site_kernels && e->P.n == nc && e->P.k == kSiteForecast && e->P.F == 2 * nc + kSiteForecast + 2 &&
           e->P.mtail_w == ((kSiteForecast + 2 + 3) & ~3);
}
// every environment is known to be inside its episode (ALIVE): whole quads, autoreset, clocks in range
bool all_alive_ok(const evc_engine* e) {
    static const bool off = getenv("EVC_ALIVE_KERNELS") && atoi(getenv("EVC_ALIVE_KERNELS")) == 0;     // measurements
    return !off && e->P.N % 4 == 0 && e->P.autoreset != 0 && e->clocks_in_range;
}

void compute_grids(evc_engine* e) {
    int blocks = (e->P.N + 3) / 4;
    int cap = 4096;
    if (const char* s = getenv("EVC_GRID_CAP")) cap = atoi(s) > 0 ? atoi(s) : cap;   // tuning knob
    if (blocks > cap) blocks = cap;
    if (blocks >= 8) blocks -= blocks % 8;
    e->step_grid = blocks;
    // Slow kernel: four queued environments per 256-thread workgroup (one per wavefront), EVC_SOLVER_WAVES workgroups
    // resident per CU.  On a quiet network the queue is empty (the launch costs ~2 us); on a congested one (most EVSEs
    // occupied, feeder limits binding) a third of the environments queue up.
    int scap = EVC_SOLVER_WAVES * e->num_cus;
    if (const char* s = getenv("EVC_SOLVER_GRID")) scap = atoi(s) > 0 ? atoi(s) : scap;
    e->solver_grid = (e->P.N + 3) / 4 < scap ? (e->P.N + 3) / 4 : scap;
    // Streaming (quad) kernel: persistent-style grid of 4 workgroups per CU (= the 4 waves/SIMD its
    // register footprint admits), every wave walks several quads.  Measured best on MI355X
    // (tools/ab_caps.py): 1024 workgroups 30-32 us vs 4096 workgroups 35.6 us per step at N = 65 536.
    int qblocks = (((e->P.N + 3) / 4) + 3) / 4;          // quads per wave, 4 waves per workgroup
    int qcap = (e->compact ? EVC_CQUAD_WAVES : EVC_QUAD_WAVES) * e->num_cus;
    if (const char* s = getenv("EVC_GRID_CAP")) qcap = atoi(s) > 0 ? atoi(s) : qcap;
    if (qblocks > qcap) qblocks = qcap;
    if (qblocks >= 8) qblocks -= qblocks % 8;
    if (qblocks < 1) qblocks = 1;
    e->quad_grid = qblocks;
    {   // the projecting lean kernels of the compact layout: EVC_PROJ_WAVES workgroups per CU
        int rb = (((e->P.N + 3) / 4) + 3) / 4;
        int rcap = EVC_PROJ_WAVES * e->num_cus;
        if (const char* s = getenv("EVC_GRID_CAP")) rcap = atoi(s) > 0 ? atoi(s) : rcap;
        if (rb > rcap) rb = rcap;
        if (rb >= 8) rb -= rb % 8;
        e->proj_grid = e->compact ? (rb < 1 ? 1 : rb) : e->quad_grid;
    }
    // the 4-environments-per-wavefront kernel evaluates constraint row c in lane c of a 16-lane
    // row and addresses whole arrays with 32-bit byte offsets
    const char* kk = getenv("EVC_KERNEL");
    const bool fits32 = (double)e->P.N * e->P.F * 4.0 < 2.0e9 && (double)e->P.N * e->P.n * 8.0 < 2.0e9 &&
                        (double)e->P.bank_slots * e->P.max_sessions * 8.0 < 2.0e9 &&
                        (double)e->P.moer_days * EVC_MOER_ROWS * EVC_MOER_COLS * 4.0 < 2.0e9;
    // the compact streaming kernel additionally needs its arrays inside one 2 GiB window
    e->use_quad = e->P.m <= 16 && fits32 && !(kk && strcmp(kk, "wave") == 0) && !(e->compact && e->P.win_span == 0u);
}

void launch_random_actions(evc_engine* e, int bins, float* actions_dev) {
    const size_t total = (size_t)e->P.N * ((e->P.n + 3) / 4);
    int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(random_actions_kernel, dim3(blocks), dim3(256), 0, e->stream, (const int4*)e->d_scal,
                       actions_dev, e->P.N, e->P.n, bins, e->policy_seed, e->env_id_base);
}

int launch_step(evc_engine* e, const void* actions_dev, int action_kind, int bins,
                const evc_step_out* out) {
    if (!out || !out->obs || !out->reward || !out->terminated)
        return fail(EVC_EINVAL, "evc_step: out->obs, out->reward, out->terminated required");
    if (action_kind != EVC_ACTION_F32 && action_kind != EVC_ACTION_DISCRETE && action_kind != EVC_ACTION_GREEDY &&
        action_kind != EVC_ACTION_RANDOM)
        return fail(EVC_EINVAL, "evc_step: unknown action_kind %d", action_kind);
    if (!actions_dev && action_kind != EVC_ACTION_GREEDY && action_kind != EVC_ACTION_RANDOM)
        return fail(EVC_EINVAL, "evc_step: actions required");
    if (action_kind == EVC_ACTION_RANDOM && bins == 1)
        return fail(EVC_EINVAL, "evc_step: random discrete actions need bins >= 2 (bins <= 0: continuous)");
    if (action_kind == EVC_ACTION_DISCRETE && bins < 2)
        return fail(EVC_EINVAL, "evc_step: discrete actions need bins >= 2");
    StepIO io;
    io.actions = action_kind == EVC_ACTION_GREEDY ? nullptr : actions_dev;
    io.action_kind = action_kind == EVC_ACTION_GREEDY ? EVC_ACTION_GREEDY : EVC_ACTION_F32;
    io.bins = 0;
    io.out = *out;
    // The staged action kinds run a kernel on the engine's stream before the step (random_actions_kernel reads the
    // environments' scalars): it must not overtake half launches still pending on the side streams.  Such steps are
    // never split themselves (below).
    // the lean projecting kernels of the compact layout have the greedy rule compiled in (step_kernel_cquad<..., GREEDY = true>)
    const bool lean_greedy = action_kind == EVC_ACTION_GREEDY && e->use_quad && e->compact && e->P.project &&
                             !(out->pilots || out->rates || out->projected || out->returns);
    if (action_kind != EVC_ACTION_F32 && !lean_greedy)
        if (int rc = join_halves(e)) return rc;
    if (action_kind == EVC_ACTION_DISCRETE) {
        const size_t count = (size_t)e->P.N * e->P.n;
        if (!e->d_act_f32) HIP_TRY(dmalloc(&e->d_act_f32, count));
        int blocks = (int)((count + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(discretize_kernel, dim3(blocks), dim3(256), 0, e->stream,
                           (const long long*)actions_dev, e->d_act_f32, count, bins);
        io.actions = e->d_act_f32;
    }
    if (action_kind == EVC_ACTION_RANDOM) {
        if (!e->d_act_f32) HIP_TRY(dmalloc(&e->d_act_f32, (size_t)e->P.N * e->P.n));
        launch_random_actions(e, bins, e->d_act_f32);
        io.actions = e->d_act_f32;
    }
    // The slow-queue counter is double-buffered: step s counts in block[s & 1] and whoever drains step s
    // clears block[(s + 1) & 1] for the next step (no per-step memset kernel).
    e->P.slow_count = e->d_slow_count + (e->step_parity & 1);
    e->P.slow_count_next = e->d_slow_count + ((e->step_parity + 1) & 1);
    e->step_parity ^= 1;
    e->P.host_qlen = e->d_qlen ? e->d_qlen + (e->step_index % kQlenRing) : nullptr;
    const int words = (e->P.G + 1) / 2;
    const bool dbg = out->pilots || out->rates || out->projected || out->returns ||
                     (action_kind == EVC_ACTION_GREEDY && !lean_greedy);
    // drain mode (see evc_engine): the lean compact streaming kernel can drain short queues itself
    // Who finishes the rows whose projection needs the iterative solver (lean compact streaming kernel):
    //   1  the workgroup that queued them, from a list of its own once its streaming work is done;
    //   0  the slow kernel (a second launch) from the global queue — also every other kernel family.
    // Default: by the time of day, from the queue lengths the kernels report (below); EVC_DRAIN=0|1 forces a mode.
    // Measured in round 3 and rejected: all four wavefronts draining the list, solving a row where it
    // stands inside the period's body, solving between two quads of the wavefront that queued it — a solve's latency
    // (~20 us: dependent loads, the projection's ladders, finish_step) is what the launch waits for in every form, and a
    // call inside the loop over quads costs the streaming path 1.4 - 4 us per step.
    bool drain = false;
    int mode = 0;
    if (e->P.project && e->use_quad && e->compact && !dbg && e->h_qlen) {
        // longest queue among the last day's reports; slots no kernel has written yet hold INT_MAX, so an engine
        // starts with the slow kernel and only drops it after a whole day of short queues (the host runs ahead of
        // the GPU by many steps: a rule that followed the last few reports switched too late on a day whose
        // queues ramp up within a few periods — 107 instead of 78 us per step on JPL's GMM days, measured)
        int recent = 0;
        const volatile int* ring = (volatile int*)e->h_qlen;
        auto reported = [&](int i) {                 // a step's queue = what its launch(es) reported (two rings: two half launches)
            const int a = ring[i], b = ring[kQlenRing + i];
            return (a == INT_MAX || b == INT_MAX) ? INT_MAX : a + b;
        };
        for (int i = 0; i < kQlenRing; i++) { const int v = reported(i); if (v > recent) recent = v; }
        const int max_queue = e->pipeline == 2 ? e->drain_max_queue_pipelined : e->drain_max_queue;
        drain = recent <= max_queue;
        if (!drain && recent != INT_MAX) {
            // A day with a congested part.  The ring is indexed by the period of the day, so the slots around this step's hold
            // the last reports for this time of day — today's behind it (as far as the GPU has come), yesterday's ahead.
            // Quiet there (one hour back, three ahead): drain in the kernel — the nights of a congested day then cost one
            // launch per step instead of two.  A wrong guess only costs speed.
            int around = 0;
            const int here = (int)(e->step_index % kQlenRing);
            for (int d = -kDrainLookBack; d <= kDrainLookAhead; d++) {
                const int v = reported((here + d + kQlenRing) % kQlenRing);
                if (v > around) around = v;
            }
            drain = around <= max_queue;
        }
        if (e->drain_override >= 0) drain = e->drain_override != 0;
        // capacity guard, AFTER the override: a workgroup's list must hold every environment it steps (a queued row that found
        // the list full would not be stepped), so a launch shape that cannot guarantee it always gets the slow kernel
        const long long quads_per_wave = (((long long)e->P.N + 3) / 4 + 4LL * e->proj_grid - 1) / (4LL * e->proj_grid);
        if (quads_per_wave * 16 > kDrainListMax) drain = false;
        mode = drain ? 1 : 0;
    }
    // Pipelined halves (evc_set_pipeline): only the lean compact streaming kernel that needs no second launch (queue drained
    // in the kernel, or no projection), on actions read straight from the caller's buffer (the discretised / random forms
    // go through one staging buffer the next step would overwrite), and only where a half still fills the grid.
    const int split_cap = e->P.project ? e->proj_grid : e->quad_grid;
    const bool split = e->pipeline == 2 && e->use_quad && e->compact && !dbg &&
                       (action_kind == EVC_ACTION_F32 || lean_greedy) && ((e->P.N + 3) / 4) / 2 >= (getenv("EVC_SPLIT_MINQ") ? atoi(getenv("EVC_SPLIT_MINQ")) : 4) * split_cap;
    // The queue's control blocks and report rings exist once per half launch.  A step that is ONE launch uses the first set;
    // where the form changes, what the other form left behind is cleared (rare: a mode or action-kind change).
    if (split != e->last_split) {
        if (int rc = join_halves(e)) return rc;
        (void)hipMemsetAsync(e->d_slow_count, 0, 4 * sizeof(int), e->stream);
        e->last_split = split;
    }
    if (!split && e->h_qlen) ((volatile int*)e->h_qlen)[kQlenRing + (e->step_index % kQlenRing)] = 0;
    if (!split)
        if (int rc = join_halves(e)) return rc;
    // A caller that enqueues per-half work on the side streams (evc_pipeline_half) relies on "after this half's previous
    // launch, before its next one".  A step that is NOT split (small batches, a staged action kind, debug outputs) is one
    // launch on the engine's stream: it is ordered after whatever the caller put on both side streams, and both side streams
    // are ordered after it below — the same contract, without the overlap (it raced on the action buffer).
    const bool guard_unsplit = e->pipeline == 2 && !split && e->halves_exposed;
    if (guard_unsplit)
        for (int h = 0; h < 2; h++) {
            HIP_TRY(hipEventRecord(e->join_ev[h], e->side[h]));
            HIP_TRY(hipStreamWaitEvent(e->stream, e->join_ev[h], 0));
        }
    auto launch_split = [&](auto kernel, auto slow_kernel, bool with_slow) {
        // Both halves wait for whatever the engine's stream still holds (the caller's actions): one event, recorded there and
        // waited for by both side streams — but only if the stream holds anything.  An idle stream (actions staged earlier,
        // a replay, a caller that synchronises by itself) needs no ordering, and the event costs: 11 us of host time per
        // step and a marker between consecutive launches of a stream (24.9 instead of 22.6 us per step, measured).
        const int nq = (e->P.N + 3) / 4, mid = (nq / 2) & ~7;
        static const int fork_mode = getenv("EVC_PIPE_FORK") ? atoi(getenv("EVC_PIPE_FORK")) : -1;   // measurements: 0 never, 1 always
        const bool fork = fork_mode >= 0 ? fork_mode != 0 : hipStreamQuery(e->stream) != hipSuccess;
        if (fork) (void)hipEventRecord(e->fork_ev, e->stream);
        // Cold start (nothing pending: the first pipelined step, or the first after a join): two launches that begin together
        // share the GPU evenly and end together, tails side by side — 32 us per step, worse than one launch — and the trains
        // need five to ten steps to drift apart.  EVC_PIPE_SKEW_US > 0 starts the second half's train that much later (a
        // one-wavefront sleep kernel) for a caller that has been free-running (its previous train had at least four steps; one
        // that joins after every step never sees it).  Measured twice on windows of 20 steps, medians of 8 interleaved runs:
        // 27.9 -> 26.7 us per step with the skew on every cold start, 27.8 -> 28.3 with this rule; long runs unchanged.  No
        // reliable gain: off by default.
        static const int skew_us = getenv("EVC_PIPE_SKEW_US") ? atoi(getenv("EVC_PIPE_SKEW_US")) : kPipeSkewUs;
        // (Round 4 also built a start that de-phases the trains BY CONSTRUCTION — half A's first launch as two quarter launches,
        // half B behind the first: 25.6 / 25.6 / 27.3 / 25.5 us per step on the driver's window against 26.3 / 25.1 / 24.7 / 25.9
        // without — no gain; the path was removed in round 5 (untested quarter grids against the drain list's capacity).)
        // "drained behind our back" is told by the host clock, not by querying the streams (two runtime calls per step, and a
        // host that is only just ahead of the GPU would find them idle again and again): a gap of more than 200 us since the
        // last pipelined step was issued is at least eight step times — the trains have run dry.
        const auto now = std::chrono::steady_clock::now();
        const double gap_us = std::chrono::duration<double, std::micro>(now - e->last_split_issue).count();
        e->last_split_issue = now;
        static const double gap_limit = getenv("EVC_PIPE_GAP_US") ? atof(getenv("EVC_PIPE_GAP_US")) : 200.0;
        const bool cold = !e->halves_pending || (skew_us > 0 && gap_us > gap_limit);
        if (cold) { e->prev_train_len = e->train_len; e->train_len = 0; }
        e->train_len++;
        for (int h = 0; h < 2; h++) {
            StepIO ioh = io;
            ioh.quad_lo = h ? mid : 0;
            ioh.quad_hi = h ? nq : mid;
            // the half's own queue: control blocks, its stretch of the list (a half queues at most its own environments), report ring
            Params Ph = e->P;
            Ph.slow_count = e->P.slow_count + 2 * h;
            Ph.slow_count_next = e->P.slow_count_next + 2 * h;
            Ph.slow_list = e->P.slow_list + (h ? mid * 4 : 0);
            Ph.host_qlen = e->P.host_qlen ? e->P.host_qlen + h * kQlenRing : nullptr;
            int grid = (ioh.quad_hi - ioh.quad_lo + 3) / 4;
            if (grid > split_cap) grid = split_cap;
            if (grid >= 8) grid -= grid % 8;
            if (fork) (void)hipStreamWaitEvent(e->side[h], e->fork_ev, 0);
            if (cold && h == 1 && skew_us > 0 && !e->timing && e->prev_train_len >= 4)
                hipLaunchKernelGGL(skew_kernel, dim3(1), dim3(64), 0, e->side[h], (unsigned)skew_us * 100u);
            if (e->timing) {
                hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, e->side[h], e->ev[h ? 4 : 0], e->ev[h ? 5 : 1], 0, Ph, ioh);
            } else {
                hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, e->side[h], Ph, ioh);
            }
            if (with_slow) {
                // the slow kernel of this half right behind it on the same stream: it runs under the other half's streaming kernel
                if (e->timing && h == 0)
                    hipExtLaunchKernelGGL(slow_kernel, dim3(e->solver_grid), dim3(256), 0, e->side[h], e->ev[2], e->ev[3], 0, Ph, ioh);
                else
                    hipLaunchKernelGGL(slow_kernel, dim3(e->solver_grid), dim3(256), 0, e->side[h], Ph, ioh);
            }
        }
        (void)hipGetLastError();                 // hipStreamQuery's hipErrorNotReady is not an error
        e->halves_pending = true;
        e->split_steps++;
        e->fork_steps += fork ? 1 : 0;
    };
    // With timing on, the two kernels carry their own start / stop events (hipExtLaunchKernel: the
    // events read the dispatch packet's begin / end timestamps, i.e. the duration a kernel trace
    // reports, without the gaps between stream operations).
    auto launch = [&](auto kernel, int grid, int block, int which) {
        if (e->timing)
            hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, e->stream, e->ev[2 * which],
                                  e->ev[2 * which + 1], 0, e->P, io);
        else
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, e->stream, e->P, io);
    };
    bool solver_ran = false;
#define EVC_LAUNCH_(KDBG, KFAST, KDRAIN, KPLAIN, GRID, LEANGRID, W)                                        \
    case W:                                                                                        \
        if (e->P.project) {                                                                        \
            if (!dbg && !e->warmed && e->use_quad && e->compact) {                                 \
                /* first lean step: run the copies this engine may use once on zero environments, so that none's \
                   first launch (code load, scratch sizing: milliseconds) lands in a later mode switch */      \
                Params pw = e->P;                                                                  \
                pw.N = 0;                                                                          \
                pw.slow_count = e->d_slow_count + 4;                                               \
                pw.slow_count_next = e->d_slow_count + 5;                                          \
                pw.host_qlen = nullptr;                                                            \
                hipLaunchKernelGGL(KDRAIN, dim3(LEANGRID), dim3(256), 0, e->stream, pw, io);       \
                hipLaunchKernelGGL(KFAST, dim3(LEANGRID), dim3(256), 0, e->stream, pw, io);        \
                e->warmed = true;                                                                  \
            }                                                                                      \
            if (e->pipeline == 2 && !e->side_warmed && !dbg && e->use_quad && e->compact) {        \
                /* a stream's hardware queue sizes its scratch at the first launch that needs it (milliseconds) */ \
                Params pw = e->P;                                                                  \
                pw.N = 0;                                                                          \
                pw.slow_count = e->d_slow_count + 4;                                               \
                pw.slow_count_next = e->d_slow_count + 5;                                          \
                pw.host_qlen = nullptr;                                                            \
                StepIO iow = io;                                                                   \
                iow.quad_lo = iow.quad_hi = 8;                                                     \
                for (int h = 0; h < 2; h++) {                                                      \
                    hipLaunchKernelGGL(KDRAIN, dim3(LEANGRID), dim3(256), 0, e->side[h], pw, iow); \
                    hipLaunchKernelGGL(KFAST, dim3(LEANGRID), dim3(256), 0, e->side[h], pw, iow);  \
                    hipLaunchKernelGGL(solver_step_kernel<W>, dim3(8), dim3(256), 0, e->side[h], pw, iow); \
                }                                                                                  \
                e->side_warmed = true;                                                             \
            }                                                                                      \
            if (dbg) launch(KDBG, GRID, 256, 0);                                                   \
            else if (split && mode == 1) launch_split(KDRAIN, solver_step_kernel<W>, false);       \
            else if (split) { launch_split(KFAST, solver_step_kernel<W>, true); solver_ran = true; } \
            else if (mode == 1) launch(KDRAIN, LEANGRID, 256, 0);                                  \
            else launch(KFAST, LEANGRID, 256, 0);                                                  \
            if (mode == 0 && !split) {                                                             \
                launch(solver_step_kernel<W>, e->solver_grid, 256, 1);                              \
                solver_ran = true;                                                                 \
            }                                                                                      \
        } else if (split) {                                                                        \
            if (!e->side_warmed) {                                                                 \
                Params pw = e->P;                                                                  \
                pw.N = 0;                                                                          \
                pw.slow_count = e->d_slow_count + 4;                                               \
                pw.slow_count_next = e->d_slow_count + 5;                                          \
                StepIO iow = io;                                                                   \
                iow.quad_lo = iow.quad_hi = 8;                                                     \
                for (int h = 0; h < 2; h++)                                                        \
                    hipLaunchKernelGGL(KPLAIN, dim3(GRID), dim3(256), 0, e->side[h], pw, iow);     \
                e->side_warmed = true;                                                             \
            }                                                                                      \
            launch_split(KPLAIN, solver_step_kernel<W>, false);                                    \
        } else {                                                                                   \
            launch(KPLAIN, GRID, 256, 0);                                                          \
        }                                                                                          \
        break;
#define EVC_LAUNCH_QUAD(W)                                                                          \
    EVC_LAUNCH_((step_kernel_quad<true, W, true>), (step_kernel_quad<true, W, false>),             \
                (step_kernel_quad<true, W, false>),                                                \
                (dbg ? step_kernel_quad<false, W, true> : step_kernel_quad<false, W, false>), e->quad_grid, e->quad_grid, W)
    // the lean kernels with the site's shape compiled in (step_kernel_cquad's NC) where Params describes exactly that shape
    auto site_shape = [&](int nc) { return site_shape_ok(e, nc); };
    // ... and with every environment known to be inside its episode (ALIVE): whole quads, autoreset, clocks in range
    const bool all_alive = all_alive_ok(e);
    // (the greedy rule's copies keep the general form: they serve GMM days under policy='greedy', 58 - 69 us per step)
#define EVC_CQ_(PROJ, W, DR, WV, GR)                                                                                                   \
    ((!(GR) && site_shape(SiteStations<W>::value))                                                                                     \
         ? (all_alive ? step_kernel_cquad<PROJ, W, false, DR, WV, false, SiteStations<W>::value, SiteStations<W>::value != 0>          \
                      : step_kernel_cquad<PROJ, W, false, DR, WV, false, SiteStations<W>::value, false>)                               \
         : step_kernel_cquad<PROJ, W, false, DR, WV, GR, 0, false>)
#define EVC_LAUNCH_CQUAD(W)                                                                         \
    EVC_LAUNCH_((step_kernel_cquad<true, W, true>),                                                                    \
                (lean_greedy ? EVC_CQ_(true, W, 0, EVC_PROJ_WAVES, true) : EVC_CQ_(true, W, 0, EVC_PROJ_WAVES, false)), \
                (lean_greedy ? EVC_CQ_(true, W, 1, EVC_PROJ_WAVES, true) : EVC_CQ_(true, W, 1, EVC_PROJ_WAVES, false)), \
                (dbg ? step_kernel_cquad<false, W, true> : EVC_CQ_(false, W, false, EVC_CQUAD_WAVES, false)), e->quad_grid, e->proj_grid, W)
#define EVC_LAUNCH_WAVE(W)                                                                          \
    EVC_LAUNCH_((step_kernel<true, W>), (step_kernel<true, W>), (step_kernel<true, W>), (step_kernel<false, W>), e->step_grid, e->step_grid, W)
#define EVC_LAUNCH_ALL(L)                                                                           \
    switch (words) {                                                                               \
        L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8)                                                    \
        default: return fail(EVC_EINVAL, "unsupported class count %d", e->P.G);                    \
    }
    if (e->use_quad && e->compact) {
        EVC_LAUNCH_ALL(EVC_LAUNCH_CQUAD)
    } else if (e->use_quad) {
        EVC_LAUNCH_ALL(EVC_LAUNCH_QUAD)
    } else {
        EVC_LAUNCH_ALL(EVC_LAUNCH_WAVE)
    }
#undef EVC_LAUNCH_ALL
#undef EVC_LAUNCH_WAVE
#undef EVC_LAUNCH_CQUAD
#undef EVC_CQ_
#undef EVC_LAUNCH_QUAD
#undef EVC_LAUNCH_
    if (e->timing) {
        e->ev_valid = true;
        e->ev_slow = solver_ran;
        e->ev_split = split;
    }
    if (guard_unsplit) {
        HIP_TRY(hipEventRecord(e->fork_ev, e->stream));
        for (int h = 0; h < 2; h++) HIP_TRY(hipStreamWaitEvent(e->side[h], e->fork_ev, 0));
    }
    HIP_TRY(hipGetLastError());
    e->env_steps += (unsigned long long)e->P.N;
    e->step_index++;
    return EVC_OK;
}

// The fused rollout (evc_rollout.h): `steps` periods of a device-resident policy in ONE launch.  Available for the
// compact layout's quad geometry, when no per-station debug output is requested.
bool fused_rollout_available(const evc_engine* e, int action_kind, int ring_len, const evc_step_out* out) {
    const bool replay = action_kind == EVC_ACTION_F32 || action_kind == EVC_ACTION_DISCRETE;
    if (action_kind != EVC_ACTION_GREEDY && action_kind != EVC_ACTION_RANDOM && !replay) return false;
    // a replayed ring is addressed with 32-bit byte offsets
    if (replay && (double)ring_len * e->P.N * e->P.n * (action_kind == EVC_ACTION_DISCRETE ? 8.0 : 4.0) >= 4.0e9) return false;
    if (!e->use_quad || !e->compact) return false;
    if (out->pilots || out->rates || out->projected) return false;
    if (const char* s = getenv("EVC_ROLLOUT_FUSED")) return atoi(s) != 0;      // measurements / tests: 0 = the loop of steps
    return true;
}

int launch_rollout(evc_engine* e, const void* actions_dev, int ring_len, int action_kind, int bins, int steps, const evc_step_out* out) {
    if (!out || !out->obs || !out->reward || !out->terminated)
        return fail(EVC_EINVAL, "evc_rollout: out->obs, out->reward, out->terminated required");
    if (action_kind == EVC_ACTION_RANDOM && bins == 1)
        return fail(EVC_EINVAL, "evc_rollout: random discrete actions need bins >= 2 (bins <= 0: continuous)");
    RolloutIO io;
    io.policy = action_kind;
    io.bins = bins;
    io.actions = actions_dev;
    io.ring_len = ring_len > 0 ? ring_len : 1;
    if (action_kind == EVC_ACTION_DISCRETE && bins < 2) return fail(EVC_EINVAL, "evc_rollout: discrete actions need bins >= 2");
    io.steps = steps;
    io.env_id_base = e->env_id_base;
    io.seed = e->policy_seed;
    io.out = *out;
    const int grid = (((e->P.N + 3) / 4) + 3) / 4;          // one quad of environments per wavefront
    // Register budget of the projecting kernels (rollout_kernel's WAVES, evc_rollout.h): which of 2 / 3 wavefronts per SIMD
    // is faster depends on how often the caller's days visit the projection branch, so the engine measures — every launch
    // carries its own begin / end events, the previous launch's duration per environment-step is read when this one is
    // issued (only if it has finished: never a wait), each setting is tried until it has two readings, then the faster one
    // is used and the other re-tried every 64th launch.  EVC_ROLLOUT_WAVES=2|3 fixes the setting.
    evc_engine::RolloutTuner& T = e->roll;
    if (T.pending >= 0 && hipEventQuery(T.ev[1]) == hipSuccess) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, T.ev[0], T.ev[1]) == hipSuccess && T.pending_work > 0.0) {
            const double per = (double)ms / T.pending_work;
            double& best = T.best[T.pending - 2];
            if (T.count[T.pending - 2]++ > 0) best = best > 0.0 && best < per ? best : per;     // the first reading of a setting includes its code load
        }
        T.pending = -1;
    }
    (void)hipGetLastError();
    int waves = 3;
    if (e->P.project) {
        static const int forced = getenv("EVC_ROLLOUT_WAVES") ? atoi(getenv("EVC_ROLLOUT_WAVES")) : 0;
        if (forced == 2 || forced == 3) waves = forced;
        else if (T.count[1] < 3) waves = 3;
        else if (T.count[0] < 3) waves = 2;
        else {
            waves = T.best[0] < T.best[1] ? 2 : 3;
            if (++T.launches % 64 == 0) waves = 5 - waves;                                  // conditions change: look at the other one again
        }
    }
    hipEvent_t ev0 = e->timing ? e->ev[0] : nullptr, ev1 = e->timing ? e->ev[1] : nullptr;
    if (!e->timing && e->P.project && T.pending < 0 && T.ev[0]) {
        ev0 = T.ev[0];
        ev1 = T.ev[1];
        T.pending = waves;
        T.pending_work = (double)e->P.N * (double)steps;
    }
    // persistent grid: the workgroups that stay resident at this register budget (4 wavefronts each, `waves` per SIMD = `waves`
    // workgroups per CU); wavefronts take their further quads from the counter, which starts behind the ones assigned by position
    int pgrid = grid;
    {
        static const int cap_env = getenv("EVC_ROLLOUT_GRID") ? atoi(getenv("EVC_ROLLOUT_GRID")) : 0;     // measurements: 0 = resident workgroups
        const int resident = cap_env > 0 ? cap_env : waves * e->num_cus;
        if (pgrid > resident) pgrid = resident;
    }
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)e->d_roll_ctr, (int)(4 * pgrid), 1, e->stream));
    io.quad_counter = e->d_roll_ctr;
    io.quad_order = nullptr;
    {   // busiest quads first (one small launch in front; EVC_ROLLOUT_ORDER=0: by index — measurements)
        static const bool ordered = !(getenv("EVC_ROLLOUT_ORDER") && atoi(getenv("EVC_ROLLOUT_ORDER")) == 0);
        if (ordered && grid > pgrid) {                 // (with every quad resident from the start there is nothing to order)
            launch_rollout_order(e->P, e->d_roll_order, e->stream);
            io.quad_order = e->d_roll_order;
        }
    }
    const int words = (e->P.G + 1) / 2;
    const bool site_alive = all_alive_ok(e) && site_shape_ok(e, words == 3 ? SiteStations<3>::value : (words == 5 ? SiteStations<5>::value : 0));
    if (!launch_rollout_kernel(e->P, io, pgrid, e->stream, ev0, ev1, waves, site_alive))
        return fail(EVC_EINVAL, "unsupported class count %d", e->P.G);
    e->last_rollout_waves = waves;
    if (e->timing) {
        e->ev_valid = true;
        e->ev_slow = false;
        e->ev_split = false;                         // ev[4], ev[5] belong to an earlier pipelined step
    }
    HIP_TRY(hipGetLastError());
    e->env_steps += (unsigned long long)e->P.N * (unsigned long long)steps;
    e->step_index += (unsigned long long)steps;
    return EVC_OK;
}

int ensure_staging(evc_engine* e) {
    if (e->d_obs) return EVC_OK;
    const size_t N = e->P.N, n = e->P.n, F = e->P.F;
    HIP_TRY(hipMalloc(&e->d_act, N * n * 8));
    HIP_TRY(dmalloc(&e->d_obs, N * F));
    HIP_TRY(dmalloc(&e->d_reward, N));
    HIP_TRY(dmalloc(&e->d_term, N));
    HIP_TRY(dmalloc(&e->d_breakdown, N * 3));
    HIP_TRY(dmalloc(&e->d_final, N * F));
    HIP_TRY(dmalloc(&e->d_pilots, N * n));
    HIP_TRY(dmalloc(&e->d_rates, N * n));
    HIP_TRY(dmalloc(&e->d_proj, N * n));
    HIP_TRY(hipMemset(e->d_final, 0, sizeof(float) * N * F));
    return EVC_OK;
}

}  // namespace

extern "C" {

const char* evc_last_error(void) { return g_err; }
int evc_abi_version(void) { return EVC_ABI_VERSION; }

int evc_create(const evc_network_desc* net, int32_t num_envs, int32_t k, uint32_t flags,
               int32_t device, int32_t bank_slots, int32_t max_sessions, int32_t moer_days,
               evc_engine** out) {
    if (!out) return fail(EVC_EINVAL, "evc_create: out is NULL");
    *out = nullptr;
    if (!net || !net->constraint_matrix || !net->phase_angles_deg || !net->magnitudes || !net->evse_kind)
        return fail(EVC_EINVAL, "evc_create: incomplete network descriptor");
    if (net->n_stations < 1 || net->n_stations > EVC_MAX_STATIONS)
        return fail(EVC_EINVAL, "evc_create: n_stations must be in [1,%d]", EVC_MAX_STATIONS);
    if (net->n_constraints < 0 || net->n_constraints > EVC_MAX_CONSTRAINTS)
        return fail(EVC_EINVAL, "evc_create: n_constraints must be in [0,%d]", EVC_MAX_CONSTRAINTS);
    if (num_envs < 1) return fail(EVC_EINVAL, "evc_create: num_envs must be >= 1");
    if (k < 1 || k > 36) return fail(EVC_EINVAL, "evc_create: moer_forecast_steps must be in [1,36]");
    if (bank_slots < 1 || max_sessions < 1 || max_sessions > EVC_MAX_SESSIONS || moer_days < 1)
        return fail(EVC_EINVAL, "evc_create: bad bank_slots/max_sessions/moer_days");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(EVC_ENODEV, "evc_create: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(EVC_ENODEV, "evc_create: device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    const int cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (prop.warpSize != 64)
        return fail(EVC_ENODEV, "evc_create: device wavefront size %d != 64 (%s)", prop.warpSize, prop.gcnArchName);

    evc_engine* e = new evc_engine();
    e->device = device;
    e->flags = flags;
    e->num_cus = cu_count;
    Params& P = e->P;
    P.N = num_envs;
    P.n = net->n_stations;
    P.m = net->n_constraints;
    P.k = k;
    P.F = 2 * P.n + k + 2;
    P.bank_slots = bank_slots;
    P.max_sessions = max_sessions;
    P.moer_days = moer_days;
    P.autoreset = (flags & EVC_FLAG_AUTORESET) ? 1 : 0;
    P.autoreset_stride = 1;
    P.project = (flags & EVC_FLAG_PROJECT_ACTION) ? 1 : 0;
    P.battery_stepwise = (flags & EVC_FLAG_BATTERY_STEPWISE) ? 1 : 0;
    {
        const char* lay = getenv("EVC_LAYOUT");          // "dense" | "compact" 
        e->compact = lay ? strcmp(lay, "compact") == 0 : kDefaultCompact;
        P.compact = e->compact ? 1 : 0;
        if (const char* sk = getenv("EVC_SITE_KERNELS")) e->site_kernels = atoi(sk) != 0;     // measurements / tests (read per engine)
    }
    NetTables T;
    int rc = build_tables(net, P, T);
    if (rc != EVC_OK) { delete e; return rc; }

    const size_t N = P.N, n = P.n;
    hipError_t err = hipSuccess;
    auto A = [&](hipError_t x) { if (err == hipSuccess) err = x; };
    // The arrays the compact streaming kernel addresses through ONE buffer descriptor (struct Win) come
    // out of one allocation, so that they always lie inside one window (256-byte aligned pieces).
    {
        size_t total = 0;
        auto piece = [&](size_t bytes) { const size_t at = total; total += (bytes + 255) & ~(size_t)255; return at; };
        const size_t o_rem = piece(sizeof(double) * N * n), o_de = piece(sizeof(int) * N * n);
        const size_t o_scal = piece(sizeof(int4) * N * 2), o_acc = piece(sizeof(double) * N * 3);
        const size_t o_sess = piece(sizeof(evc_session) * (size_t)bank_slots * max_sessions);
        const size_t o_req = piece(sizeof(double) * (size_t)bank_slots * max_sessions);
        const size_t o_hist = piece(sizeof(double) * (size_t)moer_days * EVC_MOER_ROWS);
        const size_t o_moer = piece(sizeof(float) * (size_t)moer_days * EVC_MOER_ROWS * EVC_MOER_COLS);
        const size_t o_tab = piece(sizeof(NetTables));
        // observation-ordered MOER rows for the compact streaming kernel's 16-byte accesses: [forecast 1..k | prev | timestep | 0 ...],
        // k + 2 floats padded to whole 16-byte chunks
        P.mtail_w = (P.k + 2 + 3) & ~3;
        const size_t o_mtail = piece(sizeof(float) * (size_t)moer_days * EVC_MOER_ROWS * (size_t)P.mtail_w);
        A(hipMalloc((void**)&e->d_arena, total));
        if (err == hipSuccess) {
            char* b = e->d_arena;
            e->d_rem = (double*)(b + o_rem); e->d_depest = (int*)(b + o_de); e->d_scal = (int4*)(b + o_scal);
            e->d_acc = (double*)(b + o_acc); e->d_sessions = (evc_session*)(b + o_sess);
            e->d_requested = (double*)(b + o_req); e->d_moer_hist = (double*)(b + o_hist);
            e->d_moer_obs = (float*)(b + o_moer); e->d_tables = (NetTables*)(b + o_tab);
            e->d_moer_tail = (float*)(b + o_mtail);
        }
    }
    A(dmalloc(&e->d_nsess, (size_t)bank_slots));
    A(dmalloc(&e->d_slot_moer, (size_t)bank_slots));
    A(dmalloc(&e->d_maxprofit, (size_t)bank_slots));
    A(dmalloc(&e->d_slow_count, 6));
    A(dmalloc(&e->d_slow_list, N));
    A(dmalloc(&e->d_tie, 2 * kTieSlots));
    A(dmalloc(&e->d_idbuf, 2 * N));
    A(dmalloc(&e->d_metrics, 8));
    A(hipMalloc((void**)&e->d_cq_blob, sizeof(CqBlob)));
    A(hipMalloc((void**)&e->d_roll_ctr, 64));
    A(hipMalloc((void**)&e->d_roll_order, sizeof(unsigned) * ((N + 3) / 4)));
    if (err != hipSuccess) {
        free_all(e);
        delete e;
        return fail(EVC_ENOMEM, "evc_create: hipMalloc failed: %s", hipGetErrorString(err));
    }
    A(hipMemset(e->d_rem, 0, sizeof(double) * N * n));
    A(hipMemset(e->d_depest, 0xff, sizeof(int) * N * n));
    A(hipMemset(e->d_scal, 0, sizeof(int4) * N * 2));
    A(hipMemset(e->d_acc, 0, sizeof(double) * N * 3));
    A(hipMemset(e->d_sessions, 0, sizeof(evc_session) * (size_t)bank_slots * max_sessions));
    A(hipMemset(e->d_requested, 0, sizeof(double) * (size_t)bank_slots * max_sessions));
    A(hipMemset(e->d_nsess, 0, sizeof(int) * (size_t)bank_slots));
    A(hipMemset(e->d_slot_moer, 0, sizeof(int) * (size_t)bank_slots));
    A(hipMemset(e->d_maxprofit, 0, sizeof(double) * (size_t)bank_slots));
    A(hipMemset(e->d_moer_hist, 0, sizeof(double) * (size_t)moer_days * EVC_MOER_ROWS));
    A(hipMemset(e->d_moer_obs, 0, sizeof(float) * (size_t)moer_days * EVC_MOER_ROWS * EVC_MOER_COLS));
    A(hipMemset(e->d_moer_tail, 0, sizeof(float) * (size_t)moer_days * EVC_MOER_ROWS * (size_t)P.mtail_w));
    A(hipMemset(e->d_slow_count, 0, 6 * sizeof(int)));
    A(hipMemset(e->d_tie, 0, 2 * kTieSlots * sizeof(unsigned long long)));
    A(hipMemset(e->d_slow_list, 0xff, sizeof(int) * N));
    A(copy_h2d(e->d_tables, &T, sizeof(T), e->stream));
    {
        std::vector<CqBlob> blob(1);
        build_cq_blob(P, T, blob[0]);
        A(copy_h2d(e->d_cq_blob, blob.data(), sizeof(CqBlob), e->stream));
        P.cq_blob = e->d_cq_blob;
    }
    for (auto& ev : e->ev) A(hipEventCreate(&ev));
    for (auto& ev : e->roll.ev) A(hipEventCreate(&ev));
    if (err != hipSuccess) {
        free_all(e);
        delete e;
        return fail(EVC_EHIP, "evc_create: device init failed: %s", hipGetErrorString(err));
    }
    // a fresh engine is "terminated" until the first reset: t = 288
    {
        std::vector<int4> init(N * 2);
        for (size_t i = 0; i < N; i++) {
            init[2 * i] = make_int4(EVC_EPISODE_STEPS, 0, 0, 0);
            init[2 * i + 1] = make_int4(0, kNoArrival, 0, 0);
        }
        A(copy_h2d(e->d_scal, init.data(), sizeof(int4) * N * 2, e->stream));
    }
    P.rem = e->d_rem; P.depest = e->d_depest; P.scal = e->d_scal; P.acc = e->d_acc;
    P.sessions = e->d_sessions; P.requested = e->d_requested; P.n_sessions = e->d_nsess;
    P.slot_moer_day = e->d_slot_moer; P.moer_hist = e->d_moer_hist; P.moer_obs = e->d_moer_obs;
    P.tables = e->d_tables; P.slow_count = e->d_slow_count; P.slow_list = e->d_slow_list; P.tie_counters = e->d_tie;
    {   // window over the arrays the compact streaming kernel reads through one descriptor (struct Win)
        struct Arr { const void* p; size_t bytes; unsigned* off; };
        Params& Q = e->P;
        const Arr arrs[] = {
            {Q.rem, sizeof(double) * N * n, &Q.off_rem}, {Q.depest, sizeof(int) * N * n, &Q.off_de},
            {Q.scal, sizeof(int4) * N * 2, &Q.off_scal}, {Q.acc, sizeof(double) * N * 3, &Q.off_acc},
            {Q.sessions, sizeof(evc_session) * (size_t)bank_slots * max_sessions, &Q.off_sess},
            {Q.requested, sizeof(double) * (size_t)bank_slots * max_sessions, &Q.off_req},
            {Q.moer_hist, sizeof(double) * (size_t)moer_days * EVC_MOER_ROWS, &Q.off_hist},
            {Q.moer_obs, sizeof(float) * (size_t)moer_days * EVC_MOER_ROWS * EVC_MOER_COLS, &Q.off_moer},
            {e->d_tables->timestep, sizeof(float) * EVC_MOER_ROWS, &Q.off_ts},
            {e->d_moer_tail, sizeof(float) * (size_t)moer_days * EVC_MOER_ROWS * (size_t)Q.mtail_w, &Q.off_mtail}};
        uintptr_t lo = UINTPTR_MAX, hi = 0;
        for (const Arr& a : arrs) {
            lo = std::min(lo, (uintptr_t)a.p);
            hi = std::max(hi, (uintptr_t)a.p + a.bytes);
        }
        Q.win_base = (const char*)lo;
        Q.win_span = (hi - lo) < 0x7fff0000ull ? (unsigned)(hi - lo) : 0u;
        for (const Arr& a : arrs) *a.off = (unsigned)((uintptr_t)a.p - lo);
    }
    compute_grids(e);
    // queue lengths reported to the host (drain mode); without it the slow kernel simply always runs
    if (hipHostMalloc((void**)&e->h_qlen, sizeof(int) * 2 * kQlenRing, hipHostMallocMapped) == hipSuccess) {
        for (int i = 0; i < kQlenRing; i++) { e->h_qlen[i] = INT_MAX; e->h_qlen[kQlenRing + i] = 0; }
        if (hipHostGetDevicePointer((void**)&e->d_qlen, e->h_qlen, 0) != hipSuccess) e->d_qlen = nullptr;
    } else {
        e->h_qlen = nullptr;
    }
    if (const char* s = getenv("EVC_DRAIN"))
        if (*s && strcmp(s, "auto") != 0) e->drain_override = atoi(s) != 0 ? 1 : 0;
    if (const char* s = getenv("EVC_DRAIN_MAXQ")) e->drain_max_queue = e->drain_max_queue_pipelined = atoi(s);
    *out = e;
    return EVC_OK;
}

void evc_destroy(evc_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)join_halves(e);
    (void)hipStreamSynchronize(e->stream);
    free_all(e);
    delete e;
}

int evc_set_stream(evc_engine* e, void* s) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    if ((hipStream_t)s == e->stream) return EVC_OK;
    if (e->halves_pending)                        // the stream that is left waits for the side stream; ordering the new stream
        if (int rc = bind(e)) return rc;          // after the old one is the caller's business, as it always was
    e->stream = (hipStream_t)s;
    return EVC_OK;
}

int evc_set_pipeline(evc_engine* e, int32_t halves) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    if (halves != 1 && halves != 2) return fail(EVC_EINVAL, "evc_set_pipeline: halves must be 1 or 2, got %d", halves);
    if (int rc = bind(e)) return rc;
    if (halves == 2 && !e->side_ready) {
        // each object is created once: a call that failed half way is resumed, not skipped, by the next one
        if (!e->fork_ev) HIP_TRY(hipEventCreateWithFlags(&e->fork_ev, hipEventDisableTiming));
        for (int h = 0; h < 2; h++) {
            if (!e->join_ev[h]) HIP_TRY(hipEventCreateWithFlags(&e->join_ev[h], hipEventDisableTiming));
            if (!e->side[h]) HIP_TRY(hipStreamCreateWithFlags(&e->side[h], hipStreamNonBlocking));
        }
        // a HIP stream gets its hardware queue at its first launch (milliseconds): here, not inside somebody's timed loop
        for (int h = 0; h < 2; h++) {
            hipLaunchKernelGGL(side_stream_warmup_kernel, dim3(1), dim3(64), 0, e->side[h]);
            HIP_TRY(hipStreamSynchronize(e->side[h]));
        }
        // Two HIP streams are not promised to run concurrently, and sometimes they do not: the second engine of a process
        // regularly gets a pair whose kernels run one after the other (tools/probes/side_overlap.py: two 190 us spin kernels take
        // 366 us instead of 200) — its pipelined step then takes 31 us instead of 18.  Checked here with two 60 us spin kernels that
        // report their begin and end; a second-half stream that does not overlap the first is replaced (the rejected ones are kept
        // until the search ends, so that the runtime hands out other hardware queues), at most eight times.
        {
            static const bool check = !(getenv("EVC_SIDE_OVERLAP_CHECK") && atoi(getenv("EVC_SIDE_OVERLAP_CHECK")) == 0);
            unsigned long long* const stamps = (unsigned long long*)e->d_metrics;          // 8 doubles of scratch (evc_read_metrics zeroes them before use)
            auto overlaps = [&](bool& ok) -> int {
                ok = false;
                for (int rep = 0; rep < 2 && !ok; rep++) {
                    for (int h = 0; h < 2; h++) hipLaunchKernelGGL(spin_stamp_kernel, dim3(1), dim3(64), 0, e->side[h], stamps + 2 * h, 6000u);
                    for (int h = 0; h < 2; h++) HIP_TRY(hipStreamSynchronize(e->side[h]));
                    unsigned long long t[4];
                    HIP_TRY(hipMemcpy(t, stamps, sizeof(t), hipMemcpyDeviceToHost));
                    ok = t[2] < t[1] && t[0] < t[3];               // each began before the other ended
                }
                return EVC_OK;
            };
            std::vector<hipStream_t> rejected;
            bool ok = true;
            if (check) {
                if (int rc = overlaps(ok)) return rc;
                for (int attempt = 0; !ok && attempt < 8; attempt++) {
                    rejected.push_back(e->side[1]);
                    e->side[1] = nullptr;
                    HIP_TRY(hipStreamCreateWithFlags(&e->side[1], hipStreamNonBlocking));
                    hipLaunchKernelGGL(side_stream_warmup_kernel, dim3(1), dim3(64), 0, e->side[1]);
                    HIP_TRY(hipStreamSynchronize(e->side[1]));
                    if (int rc = overlaps(ok)) return rc;
                }
            }
            for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
            e->side_streams_replaced = (int)rejected.size();
            e->side_streams_overlap = ok;
        }
        e->side_ready = true;
    }
    e->pipeline = halves;
    return EVC_OK;
}

int evc_join(evc_engine* e) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    return bind(e);
}

int evc_pipeline_half(evc_engine* e, int32_t half, void** hip_stream, int32_t* env_lo, int32_t* env_hi) {
    if (!e || !hip_stream || !env_lo || !env_hi) return fail(EVC_EINVAL, "evc_pipeline_half: null argument");
    if (half != 0 && half != 1) return fail(EVC_EINVAL, "evc_pipeline_half: half must be 0 or 1, got %d", half);
    if (!e->side_ready) return fail(EVC_ESTATE, "evc_pipeline_half: call evc_set_pipeline(e, 2) first");
    const int nq = (e->P.N + 3) / 4, mid = (nq / 2) & ~7;      // the split of launch_split
    e->halves_exposed = true;
    *hip_stream = (void*)e->side[half];
    *env_lo = half ? mid * 4 : 0;
    *env_hi = half ? e->P.N : mid * 4;
    return EVC_OK;
}

int evc_pipelined_steps(evc_engine* e, uint64_t* count, uint64_t* ordered) {
    if (!e || !count) return fail(EVC_EINVAL, "null argument");
    *count = e->split_steps;
    if (ordered) *ordered = e->fork_steps;
    return EVC_OK;
}

int evc_synchronize(evc_engine* e) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    if (int rc = bind(e)) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return EVC_OK;
}

int evc_obs_dim(const evc_engine* e) { return e ? e->P.F : EVC_EINVAL; }
int evc_num_envs(const evc_engine* e) { return e ? e->P.N : EVC_EINVAL; }
int evc_num_stations(const evc_engine* e) { return e ? e->P.n : EVC_EINVAL; }
int evc_num_groups(const evc_engine* e) { return e ? e->P.G : EVC_EINVAL; }

int evc_upload_moer(evc_engine* e, int32_t first_day, int32_t num_days, const double* moer) {
    if (!e || !moer) return fail(EVC_EINVAL, "evc_upload_moer: null argument");
    if (first_day < 0 || num_days < 1 || first_day + num_days > e->P.moer_days)
        return fail(EVC_EINVAL, "evc_upload_moer: days [%d,%d) outside capacity %d", first_day,
                    first_day + num_days, e->P.moer_days);
    if (int rc = bind(e)) return rc;
    const size_t rows = (size_t)num_days * EVC_MOER_ROWS;
    std::vector<double> hist(rows);
    std::vector<float> obs(rows * EVC_MOER_COLS);
    for (size_t r = 0; r < rows; r++) {
        hist[r] = moer[r * EVC_MOER_COLS];                       // env.py:455 uses float64 col 0
        for (int c = 0; c < EVC_MOER_COLS; c++)                  // env.py:390-391 casts to float32
            obs[r * EVC_MOER_COLS + c] = (float)moer[r * EVC_MOER_COLS + c];
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(copy_h2d(e->d_moer_hist + (size_t)first_day * EVC_MOER_ROWS, hist.data(), sizeof(double) * rows, e->stream));
    HIP_TRY(copy_h2d(e->d_moer_obs + (size_t)first_day * EVC_MOER_ROWS * EVC_MOER_COLS, obs.data(), sizeof(float) * rows * EVC_MOER_COLS, e->stream));
    {   // the same values in observation order (env.py:390-392: forecast columns 1..k, prev = column 0, timestep), one padded row per period
        const int k = e->P.k, w = e->P.mtail_w;
        std::vector<float> tail(rows * (size_t)w, 0.0f);
        for (size_t r = 0; r < rows; r++) {
            float* row = &tail[r * (size_t)w];
            for (int c = 0; c < k; c++) row[c] = obs[r * EVC_MOER_COLS + 1 + c];
            row[k] = obs[r * EVC_MOER_COLS];
            row[k + 1] = (float)((double)(r % EVC_MOER_ROWS) / (double)EVC_EPISODE_STEPS);
        }
        HIP_TRY(copy_h2d(e->d_moer_tail + (size_t)first_day * EVC_MOER_ROWS * (size_t)w, tail.data(), sizeof(float) * rows * (size_t)w, e->stream));
    }
    return EVC_OK;
}

int evc_upload_episodes(evc_engine* e, int32_t first_slot, int32_t count, int32_t stride,
                        const int32_t* n_sessions, const evc_session* sessions,
                        const double* requested, const int32_t* moer_day) {
    if (!e || !n_sessions || !sessions || !requested || !moer_day)
        return fail(EVC_EINVAL, "evc_upload_episodes: null argument");
    const Params& P = e->P;
    if (first_slot < 0 || count < 1 || first_slot + count > P.bank_slots)
        return fail(EVC_EINVAL, "evc_upload_episodes: slots [%d,%d) outside bank of %d", first_slot,
                    first_slot + count, P.bank_slots);
    if (stride < 1) return fail(EVC_EINVAL, "evc_upload_episodes: stride must be >= 1");
    std::vector<evc_session> s((size_t)count * P.max_sessions);
    std::vector<double> rq((size_t)count * P.max_sessions, 0.0);
    std::vector<double> profit((size_t)count, 0.0);
    memset(s.data(), 0, sizeof(evc_session) * s.size());
    for (int i = 0; i < count; i++) {
        const int ns = n_sessions[i];
        if (ns < 0 || ns > P.max_sessions || ns > stride)
            return fail(EVC_EINVAL, "episode %d: %d sessions exceed capacity %d", i, ns, P.max_sessions);
        if (moer_day[i] < 0 || moer_day[i] >= P.moer_days)
            return fail(EVC_EINVAL, "episode %d: moer_day %d outside [0,%d)", i, moer_day[i], P.moer_days);
        for (int j = 0; j < ns; j++) {
            const evc_session& x = sessions[(size_t)i * stride + j];
            const double r = requested[(size_t)i * stride + j];
            if (x.arrival < 0 || x.arrival > EVC_EPISODE_STEPS || x.departure < 0 ||
                x.departure > EVC_EPISODE_STEPS)
                return fail(EVC_EINVAL, "episode %d session %d: timestamps outside [0,288]", i, j);
            if (x.station < 0 || x.station >= P.n)
                return fail(EVC_EINVAL, "episode %d session %d: station %d outside [0,%d)", i, j, x.station, P.n);
            if (j > 0 && x.arrival < sessions[(size_t)i * stride + j - 1].arrival)
                return fail(EVC_EINVAL, "episode %d: sessions must be sorted by arrival", i);
            if (!(r >= 0.0) || r > Consts::BATTERY_CAPACITY)
                return fail(EVC_EINVAL, "episode %d session %d: requested energy %g outside [0,100] kWh "
                            "(requested_energy_cap > battery capacity is unsupported)", i, j, r);
            s[(size_t)i * P.max_sessions + j] = x;
            rq[(size_t)i * P.max_sessions + j] = r;
            profit[i] += std::fmin(r, (double)(x.departure - x.arrival) * 32.0 * Consts::A_PERS_TO_KWH) *
                         Consts::MARGINAL_PROFIT_PER_KWH;                              // env.py:422-429
        }
    }
    if (int rc = bind(e)) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(copy_h2d(e->d_sessions + (size_t)first_slot * P.max_sessions, s.data(), sizeof(evc_session) * s.size(), e->stream));
    HIP_TRY(copy_h2d(e->d_requested + (size_t)first_slot * P.max_sessions, rq.data(), sizeof(double) * rq.size(), e->stream));
    HIP_TRY(copy_h2d(e->d_nsess + first_slot, n_sessions, sizeof(int) * count, e->stream));
    HIP_TRY(copy_h2d(e->d_slot_moer + first_slot, moer_day, sizeof(int) * count, e->stream));
    HIP_TRY(copy_h2d(e->d_maxprofit + first_slot, profit.data(), sizeof(double) * count, e->stream));
    return EVC_OK;
}

int evc_upload_gmm(evc_engine* e, const evc_gmm_desc* g) {
    if (!e || !g || !g->cum_weights || !g->means || !g->chol || !g->daily_counts || !g->station_usage)
        return fail(EVC_EINVAL, "evc_upload_gmm: null argument");
    if (g->n_components < 1 || g->n_components > EVC_MAX_GMM_COMPONENTS)
        return fail(EVC_EINVAL, "evc_upload_gmm: %d components outside [1,%d]", g->n_components, EVC_MAX_GMM_COMPONENTS);
    if (g->n_counts < 1 || g->n_counts > EVC_MAX_DAILY_COUNTS)
        return fail(EVC_EINVAL, "evc_upload_gmm: %d daily counts outside [1,%d]", g->n_counts, EVC_MAX_DAILY_COUNTS);
    if (g->num_days < 1 || g->num_days > e->P.moer_days)
        return fail(EVC_EINVAL, "evc_upload_gmm: num_days %d outside [1, moer_days = %d]", g->num_days, e->P.moer_days);
    if (!(g->requested_energy_cap >= 0.0) || g->requested_energy_cap > Consts::BATTERY_CAPACITY)
        return fail(EVC_EINVAL, "evc_upload_gmm: requested_energy_cap %g outside [0,100] kWh", g->requested_energy_cap);
    GenTables T;
    memset(&T, 0, sizeof(T));
    T.K = g->n_components; T.n_counts = g->n_counts; T.num_days = g->num_days;
    T.cap = g->requested_energy_cap;
    for (int k = 0; k < T.K; k++) {
        T.cum[k] = g->cum_weights[k];
        if (!(T.cum[k] >= (k ? T.cum[k - 1] : 0.0)) || T.cum[k] > 1.0)
            return fail(EVC_EINVAL, "evc_upload_gmm: cum_weights must be non-decreasing in [0,1]");
    }
    memcpy(T.means, g->means, sizeof(double) * 4 * T.K);
    memcpy(T.chol, g->chol, sizeof(double) * 16 * T.K);
    for (int i = 0; i < T.n_counts; i++) {
        if (g->daily_counts[i] < 0) return fail(EVC_EINVAL, "evc_upload_gmm: negative daily count");
        T.counts[i] = g->daily_counts[i];
    }
    unsigned long long total = 0;
    for (int i = 0; i < e->P.n; i++) { T.usage[i] = g->station_usage[i]; total += T.usage[i]; }
    if (total >= (1ull << 31)) return fail(EVC_EINVAL, "evc_upload_gmm: station usage counts sum to >= 2^31");
    if (int rc = bind(e)) return rc;
    if (!e->d_gen) HIP_TRY(hipMalloc(&e->d_gen, sizeof(GenTables)));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(copy_h2d(e->d_gen, &T, sizeof(T), e->stream));
    return EVC_OK;
}

int evc_generate_episodes(evc_engine* e, int32_t first_slot, int32_t count, uint64_t seed,
                          uint64_t first_episode) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    if (!e->d_gen) return fail(EVC_ESTATE, "evc_generate_episodes: no model uploaded (evc_upload_gmm)");
    const Params& P = e->P;
    if (first_slot < 0 || count < 1 || first_slot + count > P.bank_slots)
        return fail(EVC_EINVAL, "evc_generate_episodes: slots [%d,%d) outside bank of %d", first_slot,
                    first_slot + count, P.bank_slots);
    if (int rc = bind(e)) return rc;
    const int grid = std::min(count, e->num_cus * 32);
    hipLaunchKernelGGL(generate_kernel, dim3(grid), dim3(64), 0, e->stream, e->d_gen, P.n, P.max_sessions,
                       e->d_sessions, e->d_requested, e->d_nsess, e->d_slot_moer, e->d_maxprofit, first_slot,
                       count, (unsigned long long)seed, (unsigned long long)first_episode);
    HIP_TRY(hipGetLastError());
    return EVC_OK;
}

int evc_download_episodes(evc_engine* e, int32_t first_slot, int32_t count, int32_t stride,
                          int32_t* n_sessions, evc_session* sessions, double* requested,
                          int32_t* moer_day, double* max_profit) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    const Params& P = e->P;
    if (first_slot < 0 || count < 1 || first_slot + count > P.bank_slots)
        return fail(EVC_EINVAL, "evc_download_episodes: slots [%d,%d) outside bank of %d", first_slot,
                    first_slot + count, P.bank_slots);
    if ((sessions || requested) && stride < P.max_sessions)
        return fail(EVC_EINVAL, "evc_download_episodes: stride %d < max_sessions %d", stride, P.max_sessions);
    if (int rc = bind(e)) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    const size_t S = (size_t)P.max_sessions;
    if (n_sessions) HIP_TRY(copy_d2h(n_sessions, e->d_nsess + first_slot, sizeof(int) * count, e->stream));
    if (moer_day) HIP_TRY(copy_d2h(moer_day, e->d_slot_moer + first_slot, sizeof(int) * count, e->stream));
    if (max_profit) HIP_TRY(copy_d2h(max_profit, e->d_maxprofit + first_slot, sizeof(double) * count, e->stream));
    // rows of the bank are contiguous when the caller's stride equals max_sessions (the usual case): one copy;
    // wider host rows: contiguous copy into a scratch vector, rows spread on the CPU
    auto rows_out = [&](void* dst, const void* src_dev, size_t elem) -> int {
        if ((size_t)stride == S) {
            HIP_TRY(copy_d2h(dst, src_dev, elem * S * count, e->stream));
            return EVC_OK;
        }
        std::vector<char> tmp(elem * S * count);
        HIP_TRY(copy_d2h(tmp.data(), src_dev, tmp.size(), e->stream));
        for (int i = 0; i < count; i++)
            memcpy((char*)dst + (size_t)i * stride * elem, tmp.data() + (size_t)i * S * elem, elem * S);
        return EVC_OK;
    };
    if (sessions)
        if (int rc = rows_out(sessions, e->d_sessions + first_slot * S, sizeof(evc_session))) return rc;
    if (requested)
        if (int rc = rows_out(requested, e->d_requested + first_slot * S, sizeof(double))) return rc;
    return EVC_OK;
}

int evc_set_autoreset_stride(evc_engine* e, int32_t stride) {
    if (!e || stride < 0) return fail(EVC_EINVAL, "evc_set_autoreset_stride: bad argument");
    e->P.autoreset_stride = stride % e->P.bank_slots;
    return EVC_OK;
}

int evc_reset(evc_engine* e, const int32_t* env_ids, int32_t count, const int32_t* slots,
              float* obs_dev) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    const Params& P = e->P;
    if (count < 1 || count > P.N) return fail(EVC_EINVAL, "evc_reset: count %d outside [1,%d]", count, P.N);
    if (int rc = bind(e)) return rc;
    const int* d_ids = nullptr;
    const int* d_slots = nullptr;
    if (env_ids) {
        for (int i = 0; i < count; i++)
            if (env_ids[i] < 0 || env_ids[i] >= P.N) return fail(EVC_EINVAL, "evc_reset: env id %d", env_ids[i]);
        HIP_TRY(copy_h2d(e->d_idbuf, env_ids, sizeof(int) * count, e->stream));
        d_ids = e->d_idbuf;
    }
    if (slots) {
        for (int i = 0; i < count; i++)
            if (slots[i] < 0 || slots[i] >= P.bank_slots) return fail(EVC_EINVAL, "evc_reset: slot %d", slots[i]);
        HIP_TRY(copy_h2d(e->d_idbuf + P.N, slots, sizeof(int) * count, e->stream));
        d_slots = e->d_idbuf + P.N;
    }
    hipLaunchKernelGGL(reset_kernel, dim3((count + 3) / 4), dim3(256), 0, e->stream, e->P, d_ids,
                       d_slots, count, obs_dev);
    HIP_TRY(hipGetLastError());
    if (env_ids || slots) HIP_TRY(hipStreamSynchronize(e->stream));  // host id buffers may be reused
    return EVC_OK;
}

int evc_step(evc_engine* e, const void* actions_dev, int32_t action_kind, int32_t bins,
             const evc_step_out* out) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    if (int rc = bind_device(e)) return rc;       // launch_step joins pending half launches unless this step is split as well
    return launch_step(e, actions_dev, action_kind, bins, out);
}

int evc_rollout(evc_engine* e, const void* actions_dev, int32_t action_kind, int32_t bins,
                int32_t steps, int32_t ring_len, const evc_step_out* out) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    if (steps < 1) return fail(EVC_EINVAL, "evc_rollout: steps must be >= 1");
    const bool device_policy = action_kind == EVC_ACTION_GREEDY || action_kind == EVC_ACTION_RANDOM;
    if (!device_policy && (!actions_dev || ring_len < 1))
        return fail(EVC_EINVAL, "evc_rollout: actions and ring_len >= 1 required");
    if (int rc = bind(e)) return rc;
    if (out && fused_rollout_available(e, action_kind, ring_len, out))
        return launch_rollout(e, actions_dev, ring_len, action_kind, bins, steps, out);
    const size_t elem = action_kind == EVC_ACTION_DISCRETE ? 8 : 4;
    const size_t stride = (size_t)e->P.N * e->P.n * elem;
    for (int i = 0; i < steps; i++) {
        const void* a = device_policy
            ? nullptr : (const void*)((const char*)actions_dev + (size_t)(i % ring_len) * stride);
        if (int rc = launch_step(e, a, action_kind, bins, out)) return rc;
    }
    return join_halves(e);                        // like every entry point but evc_step: complete on the engine's stream
}

int evc_last_rollout_waves(evc_engine* e, int32_t* waves) {
    if (!e || !waves) return fail(EVC_EINVAL, "null argument");
    *waves = e->last_rollout_waves;
    return EVC_OK;
}

int evc_set_policy_seed(evc_engine* e, uint64_t seed, uint32_t env_id_base) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    e->policy_seed = seed;
    e->env_id_base = env_id_base;
    return EVC_OK;
}

int evc_fill_random_actions(evc_engine* e, int32_t bins, float* actions_dev) {
    if (!e || !actions_dev) return fail(EVC_EINVAL, "evc_fill_random_actions: null argument");
    if (bins == 1) return fail(EVC_EINVAL, "evc_fill_random_actions: discrete actions need bins >= 2");
    if (int rc = bind(e)) return rc;
    launch_random_actions(e, bins, actions_dev);
    HIP_TRY(hipGetLastError());
    return EVC_OK;
}

int evc_gather_agent_obs(evc_engine* e, const float* obs_dev, const float* delayed_obs_dev,
                         float* out_dev) {
    if (!e || !obs_dev || !out_dev) return fail(EVC_EINVAL, "evc_gather_agent_obs: null argument");
    if (int rc = bind(e)) return rc;
    int blocks = e->P.N < 8192 ? e->P.N : 8192;
    const bool aligned8 = (((uintptr_t)obs_dev | (uintptr_t)delayed_obs_dev | (uintptr_t)out_dev) & 7u) == 0;
    if (e->P.F % 2 == 0 && e->P.F / 2 <= 256 && aligned8 && !getenv("EVC_GATHER_GENERIC"))
        hipLaunchKernelGGL(gather_agent_obs_pairs_kernel, dim3(blocks), dim3(256), 0, e->stream, (const float2*)obs_dev,
                           (const float2*)delayed_obs_dev, (float2*)out_dev, e->P.N, e->P.n, e->P.F);
    else
        hipLaunchKernelGGL(gather_agent_obs_kernel, dim3(blocks), dim3(256), 0, e->stream, obs_dev,
                           delayed_obs_dev, out_dev, e->P.N, e->P.n, e->P.F);
    HIP_TRY(hipGetLastError());
    return EVC_OK;
}

int evc_reset_host(evc_engine* e, const int32_t* env_ids, int32_t count, const int32_t* slots,
                   float* obs_host) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    if (int rc = bind(e)) return rc;
    if (int rc = ensure_staging(e)) return rc;
    if (int rc = evc_reset(e, env_ids, count, slots, e->d_obs)) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (obs_host)
        HIP_TRY(copy_d2h(obs_host, e->d_obs, sizeof(float) * (size_t)e->P.N * e->P.F, e->stream));
    return EVC_OK;
}

int evc_host_register(void* ptr, size_t bytes) {
    if (!ptr || bytes == 0) return fail(EVC_EINVAL, "evc_host_register: bad argument");
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return EVC_OK;
}

int evc_host_unregister(void* ptr) {
    if (ptr) HIP_TRY(hipHostUnregister(ptr));
    return EVC_OK;
}

int evc_step_host(evc_engine* e, const void* actions_host, int32_t action_kind, int32_t bins,
                  const evc_step_out* oh) {
    if (!e || !oh || (!actions_host && action_kind != EVC_ACTION_GREEDY && action_kind != EVC_ACTION_RANDOM))
        return fail(EVC_EINVAL, "evc_step_host: null argument");
    if (int rc = bind(e)) return rc;
    if (int rc = ensure_staging(e)) return rc;
    const size_t N = e->P.N, n = e->P.n, F = e->P.F;
    const size_t abytes = N * n * (action_kind == EVC_ACTION_DISCRETE ? 8 : 4);
    // Page-locked caller buffers (StepEngine's own output arrays; actions a caller registered): every transfer of the step is
    // enqueued behind the kernel and the call waits ONCE — the round-4 form synchronised after each of its six transfers
    // (~15 us of round trip each, of a 424 us step at 16 384 environments).  Pageable buffers go through the bounce copies.
    hipError_t hrc = hipSuccess;
    // Round 5 — DIRECT mode (what SB3 / RLlib drive: tens to a few thousand environments, numpy in / numpy out): with page-locked
    // output buffers the kernels read the actions from and write the outputs to HOST memory themselves, over PCIe: one launch
    // and one synchronisation per step, no DMA engine in the loop (its start-up per copy, five copies per step, was most of a
    // 66 us step at 256 environments).  profiles/r5_host_direct.txt, us per step copies -> direct: 64 envs 63 -> 35, 256: 82 -> 37,
    // 1 024: 93 -> 54, 4 096: 164 -> 112, 16 384: 419 -> 355, 65 536: 1 228 -> 1 152.  EVC_HOST_DIRECT_MAX_BYTES (bytes of
    // observations per step; 0 = never) restores the copies above a size.
    const long long direct_max = getenv("EVC_HOST_DIRECT_MAX_BYTES") ? atoll(getenv("EVC_HOST_DIRECT_MAX_BYTES")) : LLONG_MAX;
    if ((long long)(N * F * 4) <= direct_max && oh->obs && oh->reward && oh->terminated &&
        (action_kind == EVC_ACTION_F32 || action_kind == EVC_ACTION_DISCRETE || !actions_host)) {
        void* da = actions_host ? HostCopier::device_view(actions_host) : nullptr;
        if (actions_host && !da) {
            // pageable actions (a fresh numpy array every step, as SB3 hands them over): through the engine's own page-locked
            // staging block (a memcpy of N x n x 4 bytes) — the previous step has been waited for, the block is free
            if (!e->h_act && hipHostMalloc((void**)&e->h_act, N * n * 8, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); e->h_act = nullptr; }
            if (e->h_act) {
                memcpy(e->h_act, actions_host, abytes);
                da = HostCopier::device_view(e->h_act);
            }
        }
        evc_step_out od{};
        od.obs = (float*)HostCopier::device_view(oh->obs);
        od.reward = (double*)HostCopier::device_view(oh->reward);
        od.terminated = (uint8_t*)HostCopier::device_view(oh->terminated);
        od.breakdown = oh->breakdown ? (double*)HostCopier::device_view(oh->breakdown) : nullptr;
        od.final_obs = oh->final_obs ? (float*)HostCopier::device_view(oh->final_obs) : e->d_final;
        od.pilots = oh->pilots ? (double*)HostCopier::device_view(oh->pilots) : nullptr;
        od.rates = oh->rates ? (double*)HostCopier::device_view(oh->rates) : nullptr;
        od.projected = oh->projected ? (double*)HostCopier::device_view(oh->projected) : nullptr;
        const bool ok = (!actions_host || da) && od.obs && od.reward && od.terminated && (!oh->breakdown || od.breakdown) &&
                        (!oh->final_obs || od.final_obs) && (!oh->pilots || od.pilots) && (!oh->rates || od.rates) &&
                        (!oh->projected || od.projected);
        if (ok) {
            if (int rc = launch_step(e, da, action_kind, bins, &od)) return rc;
            if (int rc = join_halves(e)) return rc;
            HIP_TRY(hipStreamSynchronize(e->stream));
            return EVC_OK;
        }
    }
    if (actions_host && !HostCopier::h2d_async(e->d_act, actions_host, abytes, e->stream, hrc))
        hrc = copy_h2d(e->d_act, actions_host, abytes, e->stream);
    HIP_TRY(hrc);
    evc_step_out od;
    od.obs = e->d_obs; od.reward = e->d_reward; od.terminated = e->d_term;
    od.breakdown = e->d_breakdown; od.final_obs = e->d_final;
    od.pilots = oh->pilots ? e->d_pilots : nullptr;
    od.rates = oh->rates ? e->d_rates : nullptr;
    od.projected = oh->projected ? e->d_proj : nullptr;
    od.returns = nullptr;
    if (int rc = launch_step(e, e->d_act, action_kind, bins, &od)) return rc;
    if (int rc = join_halves(e)) return rc;       // a pipelined step leaves two half launches on the side streams
    struct Xfer { void* dst; const void* src; size_t bytes; };
    const Xfer outs[4] = {{oh->terminated, e->d_term, N}, {oh->reward, e->d_reward, sizeof(double) * N},
                          {oh->breakdown, e->d_breakdown, sizeof(double) * N * 3}, {oh->obs, e->d_obs, sizeof(float) * N * F}};
    bool later[4] = {false, false, false, false};
    for (int i = 0; i < 4; i++) {
        if (!outs[i].dst) continue;
        later[i] = !HostCopier::d2h_async(outs[i].dst, outs[i].src, outs[i].bytes, e->stream, hrc);
        HIP_TRY(hrc);
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int i = 0; i < 4; i++)
        if (later[i]) HIP_TRY(copy_d2h(outs[i].dst, outs[i].src, outs[i].bytes, e->stream));
    bool any_done = true;
    if (oh->terminated) {
        any_done = false;
        for (size_t i = 0; i < N && !any_done; i++) any_done = oh->terminated[i] != 0;
    }
    // terminal observations exist only on steps that end an episode: no N x F copy on the other 287
    if (oh->final_obs && any_done)
        HIP_TRY(copy_d2h(oh->final_obs, e->d_final, sizeof(float) * N * F, e->stream));
    if (oh->pilots) HIP_TRY(copy_d2h(oh->pilots, e->d_pilots, sizeof(double) * N * n, e->stream));
    if (oh->rates) HIP_TRY(copy_d2h(oh->rates, e->d_rates, sizeof(double) * N * n, e->stream));
    if (oh->projected) HIP_TRY(copy_d2h(oh->projected, e->d_proj, sizeof(double) * N * n, e->stream));
    return EVC_OK;
}

int evc_get_env_scalars(evc_engine* e, int32_t* out_host) {
    if (!e || !out_host) return fail(EVC_EINVAL, "null argument");
    if (int rc = bind(e)) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(copy_d2h(out_host, e->d_scal, sizeof(int4) * 2 * (size_t)e->P.N, e->stream));
    for (int i = 0; i < e->P.N; i++) out_host[(size_t)i * 8 + 6] &= kStatusMask;   // hide the entry count
    return EVC_OK;
}

int evc_set_env_scalars(evc_engine* e, const int32_t* in_host) {
    if (!e || !in_host) return fail(EVC_EINVAL, "null argument");
    if (int rc = bind(e)) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    std::vector<int> sc((size_t)e->P.N * 8);
    HIP_TRY(copy_d2h(sc.data(), e->d_scal, sizeof(int4) * 2 * (size_t)e->P.N, e->stream));
    for (int i = 0; i < e->P.N; i++) {               // the entry count (compact layout) belongs to the station state
        const int count_bits = sc[(size_t)i * 8 + 6] & ~kStatusMask;
        memcpy(&sc[(size_t)i * 8], &in_host[(size_t)i * 8], sizeof(int) * 8);
        sc[(size_t)i * 8 + 6] = (in_host[(size_t)i * 8 + 6] & kStatusMask) | count_bits;
    }
    HIP_TRY(copy_h2d(e->d_scal, sc.data(), sizeof(int4) * 2 * (size_t)e->P.N, e->stream));
    bool in_range = true;
    for (int i = 0; i < e->P.N; i++) in_range = in_range && in_host[(size_t)i * 8] >= 0 && in_host[(size_t)i * 8] < EVC_EPISODE_STEPS;
    if (in_range != e->clocks_in_range) e->warmed = e->side_warmed = false;      // other copies of the lean kernels from here on: warm them like the first
    e->clocks_in_range = in_range;
    return EVC_OK;
}

int evc_get_station_state(evc_engine* e, double* rem, int16_t* dep, int16_t* est) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    if (int rc = bind(e)) return rc;
    const size_t N = (size_t)e->P.N, n = (size_t)e->P.n, cnt = N * n;
    HIP_TRY(hipStreamSynchronize(e->stream));
    std::vector<int> de(cnt);
    HIP_TRY(copy_d2h(de.data(), e->d_depest, sizeof(int) * cnt, e->stream));
    if (!e->compact) {
        if (rem) HIP_TRY(copy_d2h(rem, e->d_rem, sizeof(double) * cnt, e->stream));
        for (size_t i = 0; i < cnt; i++) {
            if (dep) dep[i] = (int16_t)(de[i] & 0xffff);
            if (est) est[i] = (int16_t)(de[i] >> 16);
        }
        return EVC_OK;
    }
    // compact layout: scatter the entries of every environment to their stations
    std::vector<double> rc(cnt);
    std::vector<int> sc(N * 8);
    HIP_TRY(copy_d2h(rc.data(), e->d_rem, sizeof(double) * cnt, e->stream));
    HIP_TRY(copy_d2h(sc.data(), e->d_scal, sizeof(int4) * 2 * N, e->stream));
    for (size_t i = 0; i < cnt; i++) {
        if (rem) rem[i] = 0.0;
        if (dep) dep[i] = (int16_t)kEmptyDep;
        if (est) est[i] = 0;
    }
    for (size_t env = 0; env < N; env++) {
        const int A = (sc[env * 8 + 6] >> kCountShift) & 0x7f;
        for (int a = 0; a < A; a++) {
            const unsigned w = (unsigned)de[env * n + a];
            const size_t i = env * n + (size_t)entry_station(w);
            if (rem) rem[i] = rc[env * n + a];
            if (dep) dep[i] = (int16_t)entry_dep(w);
            if (est) est[i] = (int16_t)entry_est(w);
        }
    }
    return EVC_OK;
}

int evc_get_entry_rank(evc_engine* e, int16_t* rank) {
    if (!e || !rank) return fail(EVC_EINVAL, "null argument");
    if (int rc = bind(e)) return rc;
    const size_t N = (size_t)e->P.N, n = (size_t)e->P.n, cnt = N * n;
    HIP_TRY(hipStreamSynchronize(e->stream));
    std::vector<int> de(cnt);
    HIP_TRY(copy_d2h(de.data(), e->d_depest, sizeof(int) * cnt, e->stream));
    if (!e->compact) {
        for (size_t i = 0; i < cnt; i++) rank[i] = (int16_t)(de[i] & 0xffff) == (int16_t)kEmptyDep ? (int16_t)-1 : (int16_t)(i % n);
        return EVC_OK;
    }
    std::vector<int> sc(N * 8);
    HIP_TRY(copy_d2h(sc.data(), e->d_scal, sizeof(int4) * 2 * N, e->stream));
    for (size_t i = 0; i < cnt; i++) rank[i] = -1;
    for (size_t env = 0; env < N; env++) {
        const int A = (sc[env * 8 + 6] >> kCountShift) & 0x7f;
        for (int a = 0; a < A; a++) rank[env * n + (size_t)entry_station((unsigned)de[env * n + a])] = (int16_t)a;
    }
    return EVC_OK;
}

int evc_set_station_state(evc_engine* e, const double* rem, const int16_t* dep, const int16_t* est) {
    return evc_set_station_state_ranked(e, rem, dep, est, nullptr);
}

int evc_set_station_state_ranked(evc_engine* e, const double* rem, const int16_t* dep, const int16_t* est, const int16_t* rank) {
    if (!e || !rem || !dep || !est) return fail(EVC_EINVAL, "null argument");
    if (int rc = bind(e)) return rc;
    const size_t N = (size_t)e->P.N, n = (size_t)e->P.n, cnt = N * n;
    std::vector<int> de(cnt);
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (!e->compact) {
        for (size_t i = 0; i < cnt; i++) de[i] = ((int)dep[i] & 0xffff) | ((int)est[i] << 16);
        HIP_TRY(copy_h2d(e->d_rem, rem, sizeof(double) * cnt, e->stream));
        HIP_TRY(copy_h2d(e->d_depest, de.data(), sizeof(int) * cnt, e->stream));
        return EVC_OK;
    }
    std::vector<double> rc(cnt, 0.0);
    std::vector<int> sc(N * 8);
    HIP_TRY(copy_d2h(sc.data(), e->d_scal, sizeof(int4) * 2 * N, e->stream));
    std::vector<size_t> order;
    for (size_t env = 0; env < N; env++) {
        order.clear();
        for (size_t s = 0; s < n; s++) {
            const size_t i = env * n + s;
            if (dep[i] == kEmptyDep) continue;
            if (dep[i] < 0 || dep[i] > EVC_EPISODE_STEPS)
                return fail(EVC_EINVAL, "evc_set_station_state: departure %d outside [0,288]", (int)dep[i]);
            if (rank && (rank[i] < 0 || (size_t)rank[i] >= n))
                return fail(EVC_EINVAL, "evc_set_station_state_ranked: entry_rank %d of an occupied EVSE outside [0,%d)", (int)rank[i], (int)n);
            order.push_back(s);
        }
        // the environment's list: by entry_rank where given (ties by station), else in station order
        if (rank) std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return rank[env * n + a] < rank[env * n + b]; });
        int A = 0;
        for (const size_t s : order) {
            const size_t i = env * n + s;
            rc[env * n + A] = rem[i];
            de[env * n + A] = (int)pack_entry(dep[i], (int)s, est[i]);
            A++;
        }
        sc[env * 8 + 6] = (sc[env * 8 + 6] & kStatusMask) | (A << kCountShift);
    }
    HIP_TRY(copy_h2d(e->d_rem, rc.data(), sizeof(double) * cnt, e->stream));
    HIP_TRY(copy_h2d(e->d_depest, de.data(), sizeof(int) * cnt, e->stream));
    HIP_TRY(copy_h2d(e->d_scal, sc.data(), sizeof(int4) * 2 * N, e->stream));
    return EVC_OK;
}

int evc_get_breakdown(evc_engine* e, double* out_host) {
    if (!e || !out_host) return fail(EVC_EINVAL, "null argument");
    if (int rc = bind(e)) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(copy_d2h(out_host, e->d_acc, sizeof(double) * 3 * (size_t)e->P.N, e->stream));
    return EVC_OK;
}

int evc_set_breakdown(evc_engine* e, const double* in_host) {
    if (!e || !in_host) return fail(EVC_EINVAL, "null argument");
    if (int rc = bind(e)) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(copy_h2d(e->d_acc, in_host, sizeof(double) * 3 * (size_t)e->P.N, e->stream));
    return EVC_OK;
}

int evc_clear_status(evc_engine* e) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    std::vector<int> sc((size_t)e->P.N * 8);
    if (int rc = evc_get_env_scalars(e, sc.data())) return rc;
    for (int i = 0; i < e->P.N; i++) sc[(size_t)i * 8 + 6] = 0;
    return evc_set_env_scalars(e, sc.data());
}

int evc_read_metrics(evc_engine* e, double* out_host) {
    if (!e || !out_host) return fail(EVC_EINVAL, "null argument");
    if (int rc = bind(e)) return rc;
    HIP_TRY(hipMemsetAsync(e->d_metrics, 0, sizeof(double) * 8, e->stream));
    int blocks = (e->P.N + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(metrics_kernel, dim3(blocks), dim3(256), 0, e->stream, e->P, e->d_metrics);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(copy_d2h(out_host, e->d_metrics, sizeof(double) * 8, e->stream));
    out_host[3] = (double)e->env_steps;
    return EVC_OK;
}

int evc_set_tie_grid(evc_engine* e, int32_t log2_steps_per_amp) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    if (log2_steps_per_amp < 8 || log2_steps_per_amp > 44)
        return fail(EVC_EINVAL, "tie grid 2^-%d A outside 2^-8 .. 2^-44 A", log2_steps_per_amp);
    if (int rc = bind(e)) return rc;
    e->P.tie_log2 = log2_steps_per_amp;
    e->P.snap_tol = Consts::PROJ_TOL + e->P.snap_load * std::ldexp(1.0, -(log2_steps_per_amp + 1));
    return EVC_OK;
}

int evc_last_slow_count(evc_engine* e, int32_t* count) {
    if (!e || !count) return fail(EVC_EINVAL, "null argument");
    if (int rc = bind(e)) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    int both[4];
    HIP_TRY(copy_d2h(both, e->d_slow_count, 4 * sizeof(int), e->stream));
    // the control block of the most recent step is cleared by the drainer of the NEXT step; the one not
    // selected for the next step holds the last count (second pair: the second half launch of a pipelined step)
    *count = both[(e->step_parity + 1) & 1] + (e->last_split ? both[2 + ((e->step_parity + 1) & 1)] : 0);
    return EVC_OK;
}

#if defined(EVC_WG_TIMING) || defined(EVC_TIMELINE)
int evc_debug_read_tie(evc_engine* e, unsigned long long* out /* [512] */, int clear) {
    if (hipStreamSynchronize(e->stream) != hipSuccess) return -4;
    if (copy_d2h(out, e->d_tie, sizeof(unsigned long long) * 2 * kTieSlots, e->stream) != hipSuccess) return -4;
    if (clear && hipMemset(e->d_tie, 0, sizeof(unsigned long long) * 2 * kTieSlots) != hipSuccess) return -4;
    return 0;
}
int evc_debug_read_slow_list(evc_engine* e, int* out, int count) {
    if (hipStreamSynchronize(e->stream) != hipSuccess) return -4;
    return copy_d2h(out, e->d_slow_list, sizeof(int) * (size_t)count, e->stream) == hipSuccess ? 0 : -4;
}
#endif

#ifdef EVC_SOLVER_STATS
int evc_debug_solver_stats(unsigned long long* out8 /* [32] */) {
    static const unsigned long long zero[32] = {0};
    if (hipDeviceSynchronize() != hipSuccess) return -4;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(evc::g_solver_stats), sizeof(zero)) != hipSuccess) return -4;
    if (hipMemcpyToSymbol(HIP_SYMBOL(evc::g_solver_stats), zero, sizeof(zero)) != hipSuccess) return -4;
    return 0;
}
#endif

int evc_enable_timing(evc_engine* e, int32_t on) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    e->timing = on != 0;
    e->ev_valid = false;
    return EVC_OK;
}

int evc_last_step_ms(evc_engine* e, float* ms_main, float* ms_slow) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    if (!e->ev_valid) return fail(EVC_ESTATE, "evc_last_step_ms: no timed step recorded");
    if (int rc = bind(e)) return rc;
    HIP_TRY(hipEventSynchronize(e->ev_slow ? e->ev[3] : e->ev[1]));
    float a = 0.f, b = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, e->ev[0], e->ev[1]));
    if (e->ev_split) {
        // two half launches side by side: from the first begin to the last end
        HIP_TRY(hipEventSynchronize(e->ev[5]));
        float lead = 0.f, other = 0.f;                            // lead > 0: the second half began later
        HIP_TRY(hipEventElapsedTime(&lead, e->ev[0], e->ev[4]));
        HIP_TRY(hipEventElapsedTime(&other, e->ev[4], e->ev[5]));
        const float begin = lead < 0.f ? lead : 0.f;
        const float end = a > lead + other ? a : lead + other;
        a = end - begin;
    }
    if (e->ev_slow) HIP_TRY(hipEventElapsedTime(&b, e->ev[2], e->ev[3]));
    if (ms_main) *ms_main = a;
    if (ms_slow) *ms_slow = e->ev_slow ? b : 0.f;
    return EVC_OK;
}

int evc_last_half_ms(evc_engine* e, float* ms_first, float* ms_second) {
    if (!e) return fail(EVC_EINVAL, "null engine");
    if (!e->ev_valid || !e->ev_split) return fail(EVC_ESTATE, "evc_last_half_ms: the last timed step was not a pipelined one");
    if (int rc = bind(e)) return rc;
    HIP_TRY(hipEventSynchronize(e->ev[1]));
    HIP_TRY(hipEventSynchronize(e->ev[5]));
    float a = 0.f, b = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, e->ev[0], e->ev[1]));
    HIP_TRY(hipEventElapsedTime(&b, e->ev[4], e->ev[5]));
    if (ms_first) *ms_first = a;
    if (ms_second) *ms_second = b;
    return EVC_OK;
}

}  // extern "C"
