// evc_device.h — gfx950 device-side building blocks of the batched EV-charging step engine.
//
// One 64-lane wavefront simulates one environment: lane i = charging station i of the
// network (n <= 64).  Everything that is "per environment" is wave-uniform, everything that is
// "per station" lives in a lane.  Cross-station reductions are done with DPP row shifts /
// row broadcasts (fp64) or with ballot + popcount on bit-planes (integer pilot sums), never
// through memory.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "evcharge.h"

namespace evc {

// The kernels with a site's shape compiled in (step_kernel_cquad's / rollout_kernel's NC): stations per packed-word count of the two
// networks the reference ships (Caltech: 54 stations, three packed words; JPL: 52, five), and the forecast horizon they assume.
constexpr int kSiteForecast = 36;
template <int WORDS> struct SiteStations { static constexpr int value = WORDS == 3 ? 54 : (WORDS == 5 ? 52 : 0); };


// ---- constants of the reference (sustaingym/envs/evcharging/env.py:99-114), evaluated in the
// same expression order as the Python source so that they carry the same float64 values ----
struct Consts {
    static constexpr double TIMESTEP_DURATION = 5.0;       // env.py:99
    static constexpr double ACTION_SCALE_FACTOR = 32.0;    // env.py:100
    static constexpr double VOLTAGE = 208.0;               // env.py:103
    static constexpr double MARGINAL_PROFIT_PER_KWH = 0.15 * 0.20;                    // :106
    static constexpr double A_MINS_TO_KWH = (1.0 / 60.0) * (VOLTAGE / 1000.0);        // :108
    static constexpr double A_PERS_TO_KWH = A_MINS_TO_KWH * TIMESTEP_DURATION;        // :111
    static constexpr double PROFIT_FACTOR = A_PERS_TO_KWH * MARGINAL_PROFIT_PER_KWH;  // :112
    static constexpr double VIOLATION_FACTOR = A_PERS_TO_KWH * 0.001;                 // :113
    static constexpr double CARBON_COST_FACTOR = A_PERS_TO_KWH * (30.85 / 1000.0);    // :114
    // event_generation.py:60-62 + acnportal Linear2StageBattery defaults
    static constexpr double BATTERY_CAPACITY = 100.0;
    static constexpr double BATTERY_MAX_POWER = 100.0;
    static constexpr double TRANSITION_SOC = 0.8;
    static constexpr double FULLY_CHARGED_EPS = 1e-3;      // acnportal EV.fully_charged
    static constexpr double PROJ_TOL = 1e-10;              // rows within (1+tol) r count as satisfied
    static constexpr double PROJ_TOL_KKT = 1e-12;          // dual-gradient residual target of the solver
    static constexpr double PROJ_TOL_ACCEPT = 1e-9;        // accepted if the iteration budget runs out
    static constexpr int TIE_LOG2 = 16;                    // solver outputs snapped to a 2^-16 A grid (default of Params::tie_log2)
    static constexpr double TIE_OFFSET = 0.41421356237309515;  // grid offset (in grid steps), sqrt(2)-1
};

constexpr int kWave = 64;
constexpr int kEmptyDep = -1;          // departure value of an empty EVSE
constexpr int kNoArrival = 0x7fff;     // next_arrival when the session list is exhausted

// compact layout: entry word = departure (10 bits, 0..288) | station (6 bits) | est_departure (16 bits)
constexpr int kCountShift = 16;        // status word: flags in bits 0..15, entry count in bits 16..22
constexpr int kStatusMask = 0xffff;
__host__ __device__ __forceinline__ unsigned pack_entry(int dep, int station, int est) {
    return ((unsigned)dep & 0x3ffu) | ((unsigned)station << 10) | ((unsigned)est << 16);
}
__host__ __device__ __forceinline__ int entry_dep(unsigned w) { return (int)(w & 0x3ffu); }
__host__ __device__ __forceinline__ int entry_station(unsigned w) { return (int)((w >> 10) & 63u); }
__host__ __device__ __forceinline__ int entry_est(unsigned w) { return (int)w >> 16; }

// Per-environment scalars: two int4 per environment.
struct EnvScalars {
    int t, cursor, slot, moer_day;                     // int4 #0
    int n_sessions, next_arrival, status, episodes;    // int4 #1
};

// Network tables (engine-owned, read-only, staged into LDS by each workgroup).
struct NetTables {
    // M[c][g] = A[c][rep(g)] * exp(j*deg2rad(phase[rep(g)])), stored [g][c] so that lane c
    // reads consecutive addresses; only the [G][m] corner is populated.
    double Mre[EVC_MAX_GROUPS][EVC_MAX_CONSTRAINTS];
    double Mim[EVC_MAX_GROUPS][EVC_MAX_CONSTRAINTS];
    double mag[EVC_MAX_CONSTRAINTS];
    // float32 copies for the conservative (screening) row tests of the streaming kernel
    float Mre32[EVC_MAX_GROUPS][EVC_MAX_CONSTRAINTS];
    float Mim32[EVC_MAX_GROUPS][EVC_MAX_CONSTRAINTS];
    // screening thresholds (squared, float32, with safety margin):
    //   thr_y2[c]: class sums are upper bounds in 1/8 A units -> ((r_c - slack_c)(1-1e-4) * 8)^2,
    //              slack_c = sum_g |A_cg| n_g / 8 covers the quantisation of ceil(8 y)
    //   thr_p2[c]: pilots are exact integers -> (r_c (1-1e-4))^2
    float thr_y2[EVC_MAX_CONSTRAINTS];
    float thr_p2[EVC_MAX_CONSTRAINTS];
    //   thr_yp2[c]: as thr_y2 but with the slack widened by the worst-case rounding of
    //              env.py:373-378 (+0.5 A per AV station, +4 A per CC station): if the y screen
    //              passes THIS threshold the rounded pilots cannot violate row c either
    float thr_yp2[EVC_MAX_CONSTRAINTS];
    float timestep[EVC_MOER_ROWS];   // (float)((double)t / 288.0), env.py:392
};

constexpr int kCqBlobMaxChunks = 1024;     // Gp <= 16 rows x (2 x 16 + 2 x 8 chunks) + mag / thresholds + 132 station chunks < 1024
struct CqBlob {
    unsigned count;                          // chunks in use
    unsigned pad[3];
    unsigned dst[kCqBlobMaxChunks];          // byte offset of chunk i inside the workgroup's CquadLds
    uint4 data[kCqBlobMaxChunks];
};

// Kernel parameters (passed by value; lives in kernarg memory -> scalar loads).
struct Params {
    int N, n, m, G, k, F;
    int bank_slots, max_sessions, moer_days;
    int autoreset, autoreset_stride, project;
    // State layout.  compact = 0: rem / depest rows are indexed by station.  compact = 1: they hold
    // the A plugged-in EVs of the environment as entries 0..A-1 in arbitrary order (kernels touch
    // only those: traffic and work scale with the EVs present, not with the stations), depest packs
    // pack_entry(), and A sits in bits 16..22 of the status word.
    int compact;
    int battery_stepwise;    // Linear2StageBattery charge_calculation: 0 = "continuous" (acnportal default), 1 = "stepwise"
    unsigned long long group_mask[EVC_MAX_GROUPS];  // lanes of each station class
    unsigned long long cc_mask;                     // lanes with a ClipperCreek EVSE
    // "simple" rows load a single station class (e.g. the Caltech pod breakers): they cap that
    // class' sum at magnitude/|coefficient|; +inf if the class has no such row.  Violations of
    // simple rows only are projected in closed form (water-filling) inside the streaming kernel.
    double class_cap[EVC_MAX_GROUPS];
    unsigned simple_rows;                           // bit c set: row c is simple
    unsigned cap_classes;                           // bit g set: class g carries a cap (class_cap[g] finite)
    // 1 if every row's magnitude is non-decreasing in every class sum (all pairs of a row's class phasors have a
    // non-negative inner product): lowering values then never breaks a row that held before — an environment whose
    // screen left only simple rows open is settled by capping those classes, without evaluating the other rows
    int monotone_rows;
    int tie_log2;                                   // tie-snap grid: 2^-tie_log2 A (16; evc_set_tie_grid)
    double snap_load;                               // host side: worst sum_g |A_cg| n_g / magnitude_c (what one amp of snap per station adds to a row, relative)
    double snap_tol;                                // row tolerance after the tie snap: PROJ_TOL + what snapping can add to a row / cap
    unsigned long long* tie_counters;               // [kTieSlots][2] tie_snap_counted (64-bit: a long soak moves > 2^31 values)
    double prox_step;                               // 1 / (Gershgorin bound on lambda_max(B B')): step of the solver's proximal-gradient safeguard
    // persistent state
    double* rem;             // [N][n] remaining demand (kWh) of the plugged EV
    int* depest;             // [N][n] departure (low 16) | est_departure (high 16)
    int4* scal;              // [N][2]
    double* acc;             // [N][3] cumulative profit, carbon_cost, excess_charge
    // episode bank
    const evc_session* sessions;   // [bank_slots][max_sessions]
    const double* requested;       // [bank_slots][max_sessions]
    const int* n_sessions;         // [bank_slots]
    const int* slot_moer_day;      // [bank_slots]
    // MOER tables
    const double* moer_hist;       // [moer_days][289]        column 0 in float64 (reward)
    const float* moer_obs;         // [moer_days][289][37]    float32 (observation)
    const NetTables* tables;
    // window over the engine-owned arrays the compact streaming kernel reads (struct Win): base, span and
    // the byte offset of every array; win_span = 0 if they do not fit one 2 GiB window
    const char* win_base;
    unsigned win_span;
    unsigned off_rem, off_de, off_scal, off_acc, off_sess, off_req, off_hist, off_moer, off_ts;
    // Prologue image of the compact streaming kernel (evc_cquad.h, round 6): everything a workgroup used to DERIVE at its
    // start — the [G][m] corner of the network tables, the per-station class multipliers and EVSE kinds — as a list of
    // 16-byte chunks with their LDS byte offsets, built once at evc_create: the prologue is one pass of independent
    // loads and LDS writes instead of index arithmetic, 64-bit mask loops and dependent table loads.
    const struct CqBlob* cq_blob;
    unsigned off_mtail;            // float[moer_days][289][mtail_w]: observation tail [forecast 1..k | prev | timestep | 0 pad], 16-byte rows
    int mtail_w;                   // floats per row of that table: k + 2 rounded up to a multiple of 4
    // slow-path queue (environments whose projection needs the iterative solver)
    int* slow_count;               // control block {count, ticket} this step appends to
    int* slow_count_next;          // control block the drainer clears for the next step
    int* slow_list;                // [N]
    int* host_qlen;                // page-locked host word (device address): queue length of this step, for the engine's drain mode
};

struct StepIO {
    const void* actions;
    int action_kind, bins;
    evc_step_out out;
    // quads [quad_lo, quad_hi) of the batch (a quad = 4 consecutive environments); quad_hi = 0: all of them.  The engine's
    // pipelined mode steps the two halves of a batch as two launches on two streams (evc_set_pipeline); only the compact
    // streaming kernels (evc_cquad.h) read these.
    int quad_lo = 0, quad_hi = 0;
};

// ------------------------------------------------------------------------------------------
// row access through raw buffer resources: the SGPR descriptor carries the row's base and its
// size in bytes, the lane supplies a 32-bit byte offset.  Lanes past the end of the row read 0 and
// their stores are dropped by the hardware range check, so the row ops need neither 64-bit
// per-lane address arithmetic nor `lane < n` exec-mask branches.
// ------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int kRsrcWord3 = 0x00020000;   // gfx9 raw buffer, DATA_FORMAT = 32

__device__ __forceinline__ rsrc_t row_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, kRsrcWord3);
}
__device__ __forceinline__ double buf_ld_f64(rsrc_t r, unsigned off) {
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
__device__ __forceinline__ unsigned buf_ld_u32(rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0);
}
__device__ __forceinline__ float buf_ld_f32(rsrc_t r, unsigned off) {
    return __uint_as_float(buf_ld_u32(r, off));
}
__device__ __forceinline__ void buf_st_f64(rsrc_t r, unsigned off, double x) {
    v2u v;
    v.x = (unsigned)__double2loint(x);
    v.y = (unsigned)__double2hiint(x);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)off, 0, 0);
}
__device__ __forceinline__ void buf_st_u32(rsrc_t r, unsigned off, unsigned x) {
    __builtin_amdgcn_raw_buffer_store_b32(x, r, (int)off, 0, 0);
}
__device__ __forceinline__ void buf_st_f32(rsrc_t r, unsigned off, float x) {
    buf_st_u32(r, off, __float_as_uint(x));
}
__device__ __forceinline__ void buf_st_u8(rsrc_t r, unsigned off, unsigned char x) {
    __builtin_amdgcn_raw_buffer_store_b8(x, r, (int)off, 0, 0);
}
__device__ __forceinline__ v4u buf_ld_v4(rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
}
__device__ __forceinline__ void buf_st_v4(rsrc_t r, unsigned off, v4u x) {
    __builtin_amdgcn_raw_buffer_store_b128(x, r, (int)off, 0, 0);
}
__device__ __forceinline__ void buf_st_v2(rsrc_t r, unsigned off, v2u x) {
    __builtin_amdgcn_raw_buffer_store_b64(x, r, (int)off, 0, 0);
}
__device__ __forceinline__ void buf_st_i4(rsrc_t r, unsigned off, int4 x) {
    v4u v;
    v.x = (unsigned)x.x; v.y = (unsigned)x.y; v.z = (unsigned)x.z; v.w = (unsigned)x.w;
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, 0);
}

// Several arrays behind ONE buffer descriptor: `r` spans the window [base, base + span) that contains
// them all, `off` (an SGPR) is the array's byte offset inside it and travels in the instruction's scalar
// offset.  One descriptor + one SGPR per array instead of four SGPRs per array: the streaming kernels
// otherwise spill dozens of descriptor words (v_readlane reloads on every use).  The hardware range
// check is on voffset + soffset without 32-bit wrap (tools/scratch/soffset_probe.hip), so the
// out-of-range voffset trick still drops the access.
struct Win {
    rsrc_t r;
    unsigned off;
};
__device__ __forceinline__ double buf_ld_f64(Win w, unsigned off) {
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(w.r, (int)off, (int)w.off, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
__device__ __forceinline__ v2u buf_ld_v2(Win w, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b64(w.r, (int)off, (int)w.off, 0);
}
__device__ __forceinline__ v4u buf_ld_v4(Win w, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(w.r, (int)off, (int)w.off, 0);
}
__device__ __forceinline__ unsigned buf_ld_u32(Win w, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b32(w.r, (int)off, (int)w.off, 0);
}
__device__ __forceinline__ float buf_ld_f32(Win w, unsigned off) { return __uint_as_float(buf_ld_u32(w, off)); }
__device__ __forceinline__ void buf_st_f64(Win w, unsigned off, double x) {
    v2u v;
    v.x = (unsigned)__double2loint(x);
    v.y = (unsigned)__double2hiint(x);
    __builtin_amdgcn_raw_buffer_store_b64(v, w.r, (int)off, (int)w.off, 0);
}
__device__ __forceinline__ void buf_st_u32(Win w, unsigned off, unsigned x) {
    __builtin_amdgcn_raw_buffer_store_b32(x, w.r, (int)off, (int)w.off, 0);
}
__device__ __forceinline__ void buf_st_v4(Win w, unsigned off, v4u x) {
    __builtin_amdgcn_raw_buffer_store_b128(x, w.r, (int)off, (int)w.off, 0);
}

// ------------------------------------------------------------------------------------------
// wave-level primitives
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int CTRL, int ROW_MASK, bool BOUND_CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, BOUND_CTRL);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, BOUND_CTRL);
    return __hiloint2double(hi, lo);
}

// Inclusive prefix sum over the 64 lanes (Kogge-Stone inside each 16-lane row with DPP
// row_shr, then row_bcast:15 / row_bcast:31 to carry across rows).  Lane 63 holds the total.
__device__ __forceinline__ double wave_scan_f64(double v) {
    v += dpp_f64<0x111, 0xf, true>(v);   // row_shr:1
    v += dpp_f64<0x112, 0xf, true>(v);   // row_shr:2
    v += dpp_f64<0x114, 0xf, true>(v);   // row_shr:4
    v += dpp_f64<0x118, 0xf, true>(v);   // row_shr:8
    v += dpp_f64<0x142, 0xa, false>(v);  // row_bcast:15 -> rows 1,3
    v += dpp_f64<0x143, 0xc, false>(v);  // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_f64(double v) {
    return readlane_f64(wave_scan_f64(v), 63);
}

// butterfly min over the wave (ds_bpermute; only used on rare paths)
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return v;
}

template <int CTRL, int ROW_MASK, bool BOUND_CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, BOUND_CTRL);
}

// Wave total of a 32-bit integer (same DPP ladder as wave_scan_f64; 6 fused v_add_u32_dpp).
// Used with two 16-bit class sums packed per word.
__device__ __forceinline__ unsigned wave_total_u32(unsigned v) {
    v += dpp_u32<0x111, 0xf, true>(v);
    v += dpp_u32<0x112, 0xf, true>(v);
    v += dpp_u32<0x114, 0xf, true>(v);
    v += dpp_u32<0x118, 0xf, true>(v);
    v += dpp_u32<0x142, 0xa, false>(v);
    v += dpp_u32<0x143, 0xc, false>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Sum over the lanes in `mask` of a value that is an integer in [0, 2^BITS) on every lane:
// BITS ballots + popcounts, entirely on the scalar unit.
template <int BITS>
struct BitPlanes {
    unsigned long long plane[BITS];
    __device__ __forceinline__ void build(int v) {
#pragma unroll
        for (int b = 0; b < BITS; b++) plane[b] = __ballot((v >> b) & 1);
    }
    __device__ __forceinline__ int sum(unsigned long long mask) const {
        int s = 0;
#pragma unroll
        for (int b = 0; b < BITS; b++) s += __popcll(plane[b] & mask) << b;
        return s;
    }
};

// Philox4x32-10 (Salmon et al., SC'11): the counter-based random stream of the episode generator
// (evc_gen.h) and of the device-resident random policy (evc_kernels.h).
struct Philox {
    unsigned w[4];
    __device__ __forceinline__ Philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
        for (int round = 0; round < 10; round++) {
            const unsigned h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
            const unsigned h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
            c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        w[0] = c0; w[1] = c1; w[2] = c2; w[3] = c3;
    }
};

// ------------------------------------------------------------------------------------------
// per-station physics
// ------------------------------------------------------------------------------------------

// Tie snap of a solver-moved value: nearest point of the offset 2^-16 A grid,
// kept inside [0, h].
// `k` = Params::tie_log2 (16: the 2^-16 A grid; evc_set_tie_grid moves it, e.g. to 2^-40 A for the KKT certificate of the
// un-snapped solver output, tests/test_gpu_kkt_certificate.py).  ldexp by k = the multiply by 2^k, exactly (v_ldexp_f64).
__device__ __forceinline__ double tie_snap(double y, double h, int k) {
    const double s = ldexp(rint(ldexp(y, k) - Consts::TIE_OFFSET) + Consts::TIE_OFFSET, -k);
    return fmin(fmax(s, 0.0), h);
}

// Reach of the tie snap, counted (evc_read_metrics): counters[0] = values a projection solver moved (and snapped),
// counters[1] = those that sat within 1e-6 A of a rounding boundary of env.py:373-378 BEFORE the snap — the
// only ones for which an interior-point answer of the reference's accuracy (~1e-8 relative) could round to the
// other pilot.  Counted by the slow path and by the streaming kernels WITH per-station debug outputs; the lean
// streaming kernels pass no counters (the counting code alone cost them 1 us per step, measured).
// 1 / x and 1 / sqrt(x) for the Newton iterations of the projection solvers (evc_solver.h wave_cone, evc_rowcone.h quad_cone):
// v_rcp_f64 / v_rsq_f64 (~2^-26 relative) + two Newton steps, a few ulp — an IEEE divide is a dozen dependent instructions, a
// sqrt + divide two dozen, and a lone wavefront pays ~7 cycles for each (profiles/r4_prim_probe.txt: divide 75, sqrt 115,
// rcp + 2 Newton 50 cycles).  Only inside iterations whose fixed point does not depend on the last bits of a step.
__device__ __forceinline__ double newton_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double newton_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    r = r * __builtin_fma(-h * r, r, 1.5);
    r = r * __builtin_fma(-h * r, r, 1.5);
    return r;
}

constexpr int kTieSlots = 256;
__device__ __forceinline__ double tie_snap_counted(double y, double h, bool is_cc, unsigned long long* counters, int k) {
#ifdef EVC_ABL_NO_TIE_COUNT       /* ablation builds only: cost of the counting */
    counters = nullptr;
#endif
    const double eps = 1e-6;
    bool near;
    if (is_cc) {
        const double u = y * 0.125;
        near = fabs(u - floor(u) - 0.5) < eps * 0.125;
    } else {
        near = y > 6.0 - eps && (fabs(y - floor(y) - 0.5) < eps || fabs(y - 6.0) < eps);
    }
    if (counters) {
        // one atomic per call site and wavefront, spread over kTieSlots counter pairs: a congested midday moves
        // tens of thousands of values per step, and single-address atomics serialise
        const unsigned long long movers = __ballot(true), nears = __ballot(near);
        if (__lane_id() == (unsigned)__builtin_ctzll(movers)) {
            // slot from the hardware id (SIMD / pipe / CU bits of HW_REG_HW_ID), not from blockIdx: the slow path is
            // also reached through a call (evc_cquad.h), and a callee that wants workgroup ids makes its caller keep
            // them alive through the whole streaming loop
            const unsigned hw = (unsigned)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (7 << 11));
            unsigned long long* slot = counters + 2 * (hw & (kTieSlots - 1));
            atomicAdd(slot, (unsigned long long)__popcll(movers));
            if (nears) atomicAdd(slot + 1, (unsigned long long)__popcll(nears));
        }
    }
    return tie_snap(y, h, k);
}

// env.py:366-378: normalised action -> EVSE-legal pilot (A).  y = 32 * action (float64).
__device__ __forceinline__ double legal_pilot(double y, bool is_cc) {
    double av = (y >= 6.0) ? rint(y) : 0.0;          // np.round = half-to-even = v_rndne_f64
    double cc = rint(y / 8.0) * 8.0;
    return is_cc ? cc : av;
}

// acnportal Linear2StageBattery.charge + EV.charge for a battery of capacity 100 kWh, max power 100 kW,
// transition SoC 0.8, 5-minute periods, 208 V, whose headroom equals the EV's remaining demand `rem`
// (event_generation.py:173-176 with requested <= 100).  `pilot` is an EVSE-legal pilot (0 or 6..32 A) and
// must already be 0 for lanes without an EV.  Returns the actual rate (A), updates rem.
//
// charge_calculation = "continuous" (acnportal's default, which the reference gets: it passes no argument),
// `_charge`, restated from state of charge to headroom (rem = 100 (1 - soc), kw = 0.208 pilot <= 6.656 < 100):
//   pilot_dsoc = kw/1200 per period,  pilot_transition_soc = 1 - 0.002 kw  <=>  headroom rem_T = 0.2 kw
//   rem - rem_T >= kw/12 : constant-rate region for the whole period      rem' = rem - kw/12
//   0 < rem - rem_T < kw/12 : reaches the ramp-down line inside the period  rem' = rem_T exp(-(kw/12 - (rem - rem_T)) / rem_T)
//   rem <= rem_T : ramp-down region, rate proportional to the headroom      rem' = rem exp(-(kw/1200) / (0.002 kw)) = rem e^(-5/12)
//   amps = (rem - rem') * 12 * 1000 / 208
// (acnportal evaluates the same expressions on soc ~ 1; the two forms agree to ~1e-14 kWh per period).
//
// charge_calculation = "stepwise" (EVC_FLAG_BATTERY_STEPWISE, acnportal's legacy `_charge_stepwise`), division-free:
//   rate_to_full = 12 rem;  soc < 0.8 (rem > 20): P = min(kw, 100, 12 rem);  else P = min(kw, 5 rem);  rem' = rem - P/12

// exp(u) for u in [-5/12, 0]: Taylor series of degree 11 about the interval's centre (|v| <= 5/24:
// truncation 1.4e-17 relative), Horner in v.  Only +, * — identical on every lane and run.
__device__ __forceinline__ double exp_tail(double u) {
    const double v = u + 5.0 / 24.0;
    double p = 1.0 / 39916800.0;
    p = fma(p, v, 1.0 / 3628800.0);
    p = fma(p, v, 1.0 / 362880.0);
    p = fma(p, v, 1.0 / 40320.0);
    p = fma(p, v, 1.0 / 5040.0);
    p = fma(p, v, 1.0 / 720.0);
    p = fma(p, v, 1.0 / 120.0);
    p = fma(p, v, 1.0 / 24.0);
    p = fma(p, v, 1.0 / 6.0);
    p = fma(p, v, 0.5);
    p = fma(p, v, 1.0);
    p = fma(p, v, 1.0);
    return p * 0.8119363461506349;          // e^(-5/24)
}

constexpr double kExpTail = 0.6592406302004438;           // e^(-5/12)

// The part every lane executes.  `cross` is set on lanes in the crossing case, whose rem' / amps the
// caller finishes with charge_ev_cross() under a wave-uniform branch (at most one period per session).
__device__ __forceinline__ double charge_ev_main(double pilot, double& rem, bool stepwise, bool& cross) {
    const double kw = pilot * (Consts::VOLTAGE / 1000.0);
    if (stepwise) {
        const double bulk = fmin(12.0 * rem, Consts::BATTERY_MAX_POWER);
        const double limit = (rem > 20.0) ? bulk : 5.0 * rem;
        const double p = fmin(kw, limit);
        rem = rem - p * (1.0 / 12.0);
        cross = false;
        return p * (1000.0 / Consts::VOLTAGE);
    }
    const double rem_t = 0.2 * kw;
    const double d = kw * (1.0 / 12.0);
    const double x = rem - rem_t;
    const bool linear = x >= d;                          // also pilot = 0 (d = rem_t = 0): nothing moves
    cross = !linear && x > 0.0;
    const double delivered = linear ? d : rem - rem * kExpTail;
    rem = linear ? rem - d : rem * kExpTail;
    return delivered * (12.0 * 1000.0 / Consts::VOLTAGE);
}

// Crossing case: rem0 = headroom before the period.  Returns amps, sets rem.
__device__ __forceinline__ double charge_ev_cross(double pilot, double rem0, double& rem) {
    const double kw = pilot * (Consts::VOLTAGE / 1000.0);
    const double rem_t = 0.2 * kw;
    // u = -(kw/12 - (rem0 - rem_t)) / rem_t = rem0 / rem_t - 17/12, in (-5/12, 0)
    double r = __builtin_amdgcn_rcp(rem_t);              // ~2^-26 relative; two Newton steps -> double precision
    r = fma(fma(-rem_t, r, 1.0), r, r);
    r = fma(fma(-rem_t, r, 1.0), r, r);
    const double u = fmin(fmax(fma(rem0, r, -17.0 / 12.0), -5.0 / 12.0), 0.0);
    rem = rem_t * exp_tail(u);
    return (rem0 - rem) * (12.0 * 1000.0 / Consts::VOLTAGE);
}

// One-call form (slow kernel, wave-per-environment kernels): per-lane branch.
__device__ __forceinline__ double charge_ev(double pilot, double& rem, bool stepwise) {
    const double rem0 = rem;
    bool cross;
    double amps = charge_ev_main(pilot, rem, stepwise, cross);
    if (cross) amps = charge_ev_cross(pilot, rem0, rem);
    return amps;
}

}  // namespace evc
