/*
 * evcharge.h — C-ABI of the MI355X-native batched EV-charging step() engine.
 *
 * This is the drop-in boundary for the hot path named in BASELINE.json: the body of
 * SustainGym's EVChargingEnv.step()/reset() (reference: sustaingym/envs/evcharging/env.py)
 * over a batch of N independent environment instances, executed by hand-written gfx950
 * HIP kernels.  The reference has no FFI boundary of its own (it is pure Python calling
 * acnportal + cvxpy); each entry point below therefore cites the reference *Python*
 * interface it replaces, and INTEGRATION.md shows the ctypes stub a reference maintainer
 * would add.
 *
 * Conventions
 *   - plain C, no exceptions, no callbacks; every function returns 0 on success or a
 *     negative EVC_E* code; evc_last_error() returns a thread-local message.
 *   - the engine owns all persistent simulator state (device memory) behind an opaque
 *     handle; the caller owns every buffer it passes in.
 *   - "_dev" pointers are device (HBM) addresses valid on the engine's device; all other
 *     pointers are host addresses.  All GPU work is ordered on the engine's stream
 *     (evc_set_stream); calls taking only device pointers are asynchronous.
 *   - one host thread per handle at a time.
 *   - there is NO CPU execution path in this library: without a visible gfx950 device
 *     evc_create fails with EVC_ENODEV.
 *
 * Data layout
 *   observation row (float32[F], F = 2n + k + 2), key order = gymnasium.spaces.flatten of
 *   the reference's Dict space (keys sorted):  env.py:143-150, multiagent_env.py:88,115
 *       [0,n)        demands          (kWh)
 *       [n,2n)       est_departures   (periods)
 *       [2n,2n+k)    forecasted_moer
 *       [2n+k]       prev_moer
 *       [2n+k+1]     timestep         (t/288)
 */
#ifndef EVCHARGE_H
#define EVCHARGE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVC_ABI_VERSION 7

#define EVC_MAX_STATIONS     64   /* one gfx950 wavefront per environment            */
#define EVC_MAX_CONSTRAINTS  32   /* rows of ChargingNetwork.constraint_matrix       */
#define EVC_MAX_GROUPS       16   /* distinct (column, phase) classes of stations    */
#define EVC_MOER_ROWS        289  /* MOERLoader.retrieve -> [289,37]                 */
#define EVC_MOER_COLS        37
#define EVC_EPISODE_STEPS    288  /* env.py:124 max_timestep                         */
#define EVC_MAX_SESSIONS     256  /* per episode (real traces <= 81, GMM <= 84)      */
#define EVC_MAX_GMM_COMPONENTS 32  /* mixture components of the episode generator     */
#define EVC_MAX_DAILY_COUNTS 512   /* empirical sessions-per-day table                */
#define EVC_MAX_GENERATED_SESSIONS 128 /* per generated episode                        */

/* EVSE kinds (env.py:371-378; acnportal AeroVironment / ClipperCreek FiniteRatesEVSE) */
#define EVC_EVSE_AV 0   /* allowable pilots {0} U {6,7,...,32} A */
#define EVC_EVSE_CC 1   /* allowable pilots {0,8,16,24,32} A     */

/* evc_create flags */
#define EVC_FLAG_PROJECT_ACTION  (1u << 0)  /* env.py:118 project_action_in_env (default True) */
#define EVC_FLAG_AUTORESET       (1u << 1)  /* gymnasium 0.28 VectorEnv autoreset semantics     */
/* Battery model = acnportal Linear2StageBattery(charge_calculation=...).  The reference constructs the
 * battery without that argument (event_generation.py:173-176), i.e. with acnportal's default
 * "continuous" (maximum rate falls linearly above the transition SoC, integrated exactly over the
 * period): that is what the engine simulates unless this flag selects the legacy "stepwise" model
 * (rate limit frozen at its start-of-period value). */
#define EVC_FLAG_BATTERY_STEPWISE (1u << 2)

/* action encodings for evc_step */
#define EVC_ACTION_F32       0   /* float32[N][n] in [0,1]            (env.py:171-172)          */
#define EVC_ACTION_DISCRETE  1   /* int64[N][n] in {0..bins-1}        (wrappers.py:43-45)       */
#define EVC_ACTION_GREEDY    2   /* no action buffer: device-resident GreedyAlgorithm policy,
                                    a = 1 where observation['demands'] > 0 else 0
                                    (algorithms/evcharging/baselines.py:32-35)                   */
#define EVC_ACTION_RANDOM    3   /* no action buffer: device-resident RandomAlgorithm policy
                                    (algorithms/evcharging/baselines.py:38-51): uniform actions
                                    from a counter-based stream keyed by evc_set_policy_seed;
                                    bins >= 2 draws DiscreteActionWrapper levels instead        */

/* per-environment status bits (replace the reference's exceptions / pdb) */
#define EVC_STATUS_OCCUPIED     (1u << 0)  /* plug-in into an occupied EVSE (acnportal StationOccupiedError); session skipped */
#define EVC_STATUS_PROJ_NOCONV  (1u << 1)  /* projection did not reach the KKT tolerance     */
#define EVC_STATUS_STEP_AFTER_DONE (1u << 2) /* step() on a finished episode without autoreset; step ignored */
#define EVC_STATUS_ACTION_CLAMPED (1u << 3)  /* action outside [0,1] (or NaN) was clamped       */

/* error codes */
#define EVC_OK        0
#define EVC_EINVAL   -1
#define EVC_ENODEV   -2
#define EVC_ENOMEM   -3
#define EVC_EHIP     -4
#define EVC_ESTATE   -5

typedef struct evc_engine evc_engine;

/* Charging-network descriptor = the fields of acnportal's ChargingNetwork the reference
 * reads: cn.station_ids order (env.py:133-134), cn.constraint_matrix / cn._phase_angles /
 * cn.magnitudes (env.py:485-493, 450-451), cn.min_pilot_signals (env.py:373). */
typedef struct evc_network_desc {
    int32_t        n_stations;          /* n <= EVC_MAX_STATIONS                       */
    int32_t        n_constraints;       /* m <= EVC_MAX_CONSTRAINTS                    */
    const double*  constraint_matrix;   /* [m][n] row-major, real                      */
    const double*  phase_angles_deg;    /* [n]                                         */
    const double*  magnitudes;          /* [m] amps                                    */
    const uint8_t* evse_kind;           /* [n] EVC_EVSE_*                              */
} evc_network_desc;

/* One charging session of an episode (one acns.PluginEvent + EV, event_generation.py:167-191).
 * Timestamps are 5-minute period indices in [0,288).  Sessions of an episode are stored
 * sorted by arrival (stable). */
typedef struct evc_session {
    int16_t arrival;
    int16_t departure;
    int16_t est_departure;
    int16_t station;                    /* index into station_ids                      */
} evc_session;

/* Device output buffers of one step (any pointer except obs/reward/terminated may be NULL). */
typedef struct evc_step_out {
    float*   obs;          /* [N][F]   observation AFTER the step (after autoreset: first obs of next episode) */
    double*  reward;       /* [N]      env.py:457                                                  */
    uint8_t* terminated;   /* [N]      env.py:283 (event queue empty)                              */
    double*  breakdown;    /* [N][3]   cumulative profit, carbon_cost, excess_charge (env.py:460-462) */
    float*   final_obs;    /* [N][F]   terminal observation, written only for terminated rows      */
    double*  pilots;       /* [N][n]   pilot signals sent to the simulator (A), debug/parity       */
    double*  rates;        /* [N][n]   actual charging rates (A) = simulator.charging_rates[:,t-1] */
    double*  projected;    /* [N][n]   projected normalised action (env.py:220), debug/parity      */
    double*  returns;      /* [N]      optional accumulator: returns[i] += reward (episode return of
                                       BaseAlgorithm.run, algorithms/base.py:63-88); caller zeroes it */
} evc_step_out;

/* ---- lifecycle ------------------------------------------------------------------- */

/* Replaces EVChargingEnv.__init__ (env.py:116-176) for a batch of num_envs instances.
 * bank_slots   = number of episode slots resident in HBM (>= 1)
 * max_sessions = per-slot session capacity (<= EVC_MAX_SESSIONS)
 * moer_days    = number of [289][37] MOER day matrices resident in HBM */
int evc_create(const evc_network_desc* net, int32_t num_envs, int32_t moer_forecast_steps,
               uint32_t flags, int32_t device, int32_t bank_slots, int32_t max_sessions,
               int32_t moer_days, evc_engine** out);

/* Replaces EVChargingEnv.close (env.py:466-470). */
void evc_destroy(evc_engine* e);

const char* evc_last_error(void);
int evc_abi_version(void);

/* Orders all subsequent engine work on `hip_stream` (a hipStream_t; NULL = default). */
int evc_set_stream(evc_engine* e, void* hip_stream);
int evc_synchronize(evc_engine* e);

/* Pipelined halves (throughput loops; no counterpart in the reference, whose vector env is a set of processes).
 * halves = 2: evc_step on device buffers with the lean outputs (obs / reward / terminated / breakdown / final_obs), float32
 * actions and at least 2 x 4 quads per wavefront of the grid steps the batch as TWO launches — environments [0, N/2) and
 * [N/2, N) — on two internal streams.  Both wait for the work the engine's stream holds when evc_step is called (the
 * caller's actions: one event, skipped when that stream is idle) and consecutive steps of one half are ordered; the two
 * halves are NOT ordered with each other and
 * the engine's stream does not wait for them: a launch's tail — the few wavefronts still in the rare projection branch —
 * runs under the other half's next launch (2 x 32 768 environments: 22.6 instead of 26.4 us per 65 536 environment-steps
 * on MI355X).  The price is the contract: outputs are complete on the engine's stream only after evc_join (every other
 * entry point joins first), and action buffers must stay untouched until then.  Steps that do not qualify (per-station
 * debug outputs, discrete / random actions, a queue that needs the slow kernel, small batches) join and run as one
 * launch.  halves = 1 (default): every call is ordered on the engine's stream. */
int evc_set_pipeline(evc_engine* e, int32_t halves);
/* The engine's stream waits for the pending half launches (no-op when there are none). */
int evc_join(evc_engine* e);
/* Closed loop under the pipelined mode (a policy that READS the observation of step k to produce the action of step
 * k + 1, train_stable_baselines.py:271-275): half h (0 | 1) steps environments [*env_lo, *env_hi) on the internal stream
 * *hip_stream.  Work the caller enqueues on THAT stream — its policy for those environments: reads rows
 * [env_lo, env_hi) of the previous step's outputs, writes the same rows of the action buffer — is ordered after the
 * half's previous launch and before its next one by the stream itself, so the policy of one half runs under the other
 * half's step and no join is needed between steps.  Rows of the other half must not be touched from this stream.
 * A step the engine does not split (small batches, staged action kinds, debug outputs: one launch on the engine's stream)
 * keeps the same contract once this function has been called: it waits for both side streams and both wait for it.
 * Valid after evc_set_pipeline(e, 2); the streams live as long as the engine. */
int evc_pipeline_half(evc_engine* e, int32_t half, void** hip_stream, int32_t* env_lo, int32_t* env_hi);
/* How many steps of this engine ran as two half launches so far, and (ordered, may be NULL) how many of those found
 * work pending on the engine's stream and were ordered behind it with an event (diagnostics, tests). */
int evc_pipelined_steps(evc_engine* e, uint64_t* count, uint64_t* ordered);

/* Geometry queries. */
int evc_obs_dim(const evc_engine* e);        /* F = 2n + k + 2 */
int evc_num_envs(const evc_engine* e);
int evc_num_stations(const evc_engine* e);
int evc_num_groups(const evc_engine* e);

/* ---- episode data (reset-time inputs) --------------------------------------------- */

/* Uploads `num_days` MOER matrices (host float64 [num_days][289][37], the return value of
 * AbstractTraceGenerator.get_moer, event_generation.py:209-218) into day slots
 * [first_day, first_day+num_days). */
int evc_upload_moer(evc_engine* e, int32_t first_day, int32_t num_days, const double* moer);

/* Uploads `count` episodes into bank slots [first_slot, first_slot+count).  Replaces
 * AbstractTraceGenerator.get_event_queue (event_generation.py:149-207):
 *   n_sessions[count]; sessions[count][stride]; requested_kwh[count][stride] (already
 *   capped, event_generation.py:169-170); moer_day[count] = MOER day slot of the episode. */
int evc_upload_episodes(evc_engine* e, int32_t first_slot, int32_t count, int32_t stride,
                        const int32_t* n_sessions, const evc_session* sessions,
                        const double* requested_kwh, const int32_t* moer_day);

/* Autoreset walks the bank: next_slot = (slot + stride) mod bank_slots. */
int evc_set_autoreset_stride(evc_engine* e, int32_t stride);

/* ---- on-device episode generation (GMMsTraceGenerator on the GPU) ------------------ */

/* The model GMMsTraceGenerator samples from (event_generation.py:372-515): a K-component
 * Gaussian mixture over (arrival, departure, estimated departure [fractions of a day],
 * requested energy / 100 kWh), the empirical daily session counts, and the historical usage
 * count of every EVSE.  Host arrays, copied by evc_upload_gmm. */
typedef struct evc_gmm_desc {
    int32_t         n_components;         /* K <= EVC_MAX_GMM_COMPONENTS                  */
    int32_t         n_counts;             /* <= EVC_MAX_DAILY_COUNTS                      */
    int32_t         num_days;             /* episode day ~ U{0..num_days-1} (:117-119); <= moer_days */
    int32_t         reserved;
    const double*   cum_weights;          /* [K] cumulative mixture weights, last = 1     */
    const double*   means;                /* [K][4]                                       */
    const double*   chol;                 /* [K][4][4] lower Cholesky factor of each covariance, row-major */
    const int32_t*  daily_counts;         /* [n_counts] (:479)                            */
    const uint32_t* station_usage;        /* [n] (:497), sum < 2^31                       */
    double          requested_energy_cap; /* kWh, <= 100 (:169-170)                       */
} evc_gmm_desc;
int evc_upload_gmm(evc_engine* e, const evc_gmm_desc* gmm);

/* Generates `count` episodes into bank slots [first_slot, first_slot+count) on the engine's
 * stream (asynchronous, no host work): GMMsTraceGenerator._create_events
 * (event_generation.py:465-515) with the reference's numpy/sklearn random streams replaced by
 * the counter-based stream Philox4x32-10(key = seed, counter = (index, purpose, episode)),
 * episode = first_episode + i.  The episodes follow the reference generator's distribution;
 * they are bit-reproducible from (seed, episode) and independent of launch geometry. */
int evc_generate_episodes(evc_engine* e, int32_t first_slot, int32_t count, uint64_t seed,
                          uint64_t first_episode);

/* Reads bank slots back to the host (any argument may be NULL): n_sessions[count],
 * sessions[count][stride], requested_kwh[count][stride], moer_day[count], and
 * max_profit[count] (env.py:422-429; computed for uploaded and generated episodes alike). */
int evc_download_episodes(evc_engine* e, int32_t first_slot, int32_t count, int32_t stride,
                          int32_t* n_sessions, evc_session* sessions, double* requested_kwh,
                          int32_t* moer_day, double* max_profit);

/* ---- hot path ---------------------------------------------------------------------- */

/* Replaces EVChargingEnv.reset (env.py:293-338) for `count` environments.
 * env_ids (host, NULL = all envs 0..count-1), slots (host, bank slot per env, NULL = env id
 * mod bank_slots).  Writes the reset observation rows into obs_dev ([N][F], may be NULL). */
int evc_reset(evc_engine* e, const int32_t* env_ids, int32_t count, const int32_t* slots,
              float* obs_dev);

/* Replaces EVChargingEnv.step (env.py:229-291) incl. _to_schedule / _project_action /
 * acnsim.Simulator.step / _get_observation / _get_reward, for all N environments.
 * actions_dev: device [N][n]; action_kind EVC_ACTION_*; bins used for DISCRETE. */
int evc_step(evc_engine* e, const void* actions_dev, int32_t action_kind, int32_t bins,
             const evc_step_out* out);

/* `steps` consecutive evc_step calls without returning to the host in between (device-resident policy
 * or pre-staged actions): step i reads its actions at actions_dev + (i mod ring_len) * N * n elements
 * (ignored for EVC_ACTION_GREEDY / _RANDOM).  Outputs are those of the LAST step (obs, reward, terminated,
 * breakdown; final_obs of episodes that ended inside the call); out->returns accumulates every reward.
 * Counterpart of the episode loop of BaseAlgorithm.run (algorithms/base.py:63-88).
 * On the default (compact) state layout, with no per-station debug output requested, the whole call is
 * ONE kernel launch (csrc/evc_rollout.h): a wavefront keeps its four environments' state in registers
 * for all `steps` periods, draws / reads the actions itself, and writes state and outputs once at the
 * end — results identical to the loop of evc_step calls (tests/test_gpu_rollout.py).  EVC_ROLLOUT_FUSED=0
 * in the environment selects the loop (measurements). */
int evc_rollout(evc_engine* e, const void* actions_dev, int32_t action_kind, int32_t bins,
                int32_t steps, int32_t ring_len, const evc_step_out* out);

/* Diagnostics: the register budget (2 or 3 wavefronts per SIMD) the last fused evc_rollout launch ran with; 0 if none has.
 * The projecting rollout kernels exist at both and the engine keeps the one that is faster on the caller's workload (it times
 * its own launches; EVC_ROLLOUT_WAVES=2|3 fixes the choice). */
int evc_last_rollout_waves(evc_engine* e, int32_t* waves);

/* Seeds the device-resident random policy (EVC_ACTION_RANDOM).  The action of station s of environment
 * e in period t of its episode number p is a pure function of (seed, env_id_base + e, p, t, s):
 * Philox4x32-10, key = seed, counter = (t | (s/4) << 16, p, env_id_base + e, 0x504f4c43), word s%4;
 * continuous a = (w >> 8) * 2^-24, discrete level = (w * bins) >> 32.  env_id_base = global id of this
 * engine's environment 0 (multi-GPU shards draw disjoint streams).  Replaces RandomAlgorithm's
 * np.random.default_rng() (baselines.py:42), whose sequential stream cannot be drawn in parallel. */
int evc_set_policy_seed(evc_engine* e, uint64_t seed, uint32_t env_id_base);

/* Writes the actions EVC_ACTION_RANDOM would apply to the environments in their current state into
 * actions_dev (float32 [N][n]); evc_step(EVC_ACTION_RANDOM) = this + evc_step(EVC_ACTION_F32). */
int evc_fill_random_actions(evc_engine* e, int32_t bins, float* actions_dev);

/* Per-agent observations of the multi-agent environment (multiagent_env.py:102-148) for the whole
 * batch: out_dev[N][n][F].  Agent a of environment e receives the flattened observation obs_dev[e]
 * in which, when delayed_obs_dev is not NULL (periods_delay > 0, documented semantics), the
 * demands / est_departures of the OTHER agents are taken from delayed_obs_dev[e] (the observation
 * periods_delay steps ago) while its own entries, the MOER part and the timestep are current.
 * With delayed_obs_dev == NULL every agent row is a copy of obs_dev[e] (periods_delay = 0, and the
 * reference's effective behaviour for any delay). */
int evc_gather_agent_obs(evc_engine* e, const float* obs_dev, const float* delayed_obs_dev,
                         float* out_dev);

/* Host-buffer convenience variants (synchronous; staged through pinned memory).  Any
 * output pointer may be NULL. */
/* Page-locks (hipHostRegister) caller-owned host buffers used with the *_host entry points: the
 * device<->host copies of evc_step_host / evc_reset_host then run at PCIe speed (pageable buffers work
 * too, several times slower).  Unregister before the memory is freed. */
int evc_host_register(void* ptr, size_t bytes);
int evc_host_unregister(void* ptr);

int evc_reset_host(evc_engine* e, const int32_t* env_ids, int32_t count, const int32_t* slots,
                   float* obs_host);
int evc_step_host(evc_engine* e, const void* actions_host, int32_t action_kind, int32_t bins,
                  const evc_step_out* out_host);

/* ---- state access (checkpoint / metrics / tests) ----------------------------------- */

/* Per-env int32 scalars, host [N][8]: t, cursor, slot, moer_day, n_sessions, next_arrival,
 * status, episodes_done. */
int evc_get_env_scalars(evc_engine* e, int32_t* out_host);
/* Per-station state, host: remaining_kwh [N][n] float64; departure/est_departure [N][n]
 * int16 (departure == -1: EVSE empty). */
int evc_get_station_state(evc_engine* e, double* remaining_kwh, int16_t* departure,
                          int16_t* est_departure);
int evc_set_station_state(evc_engine* e, const double* remaining_kwh, const int16_t* departure,
                          const int16_t* est_departure);
/* ABI 7.  The compact layout keeps the plugged-in EVs of an environment as a LIST and the streaming
 * kernel sums the delivered amps of env.py:445 in list order, so the last bit of a reward depends on that order (the
 * history of plug-ins).  entry_rank [N][n] int16: position of the station's EV in its environment's list, -1 = EVSE
 * empty (dense layout: the station index where occupied).  evc_set_station_state_ranked rebuilds the lists in that
 * order (entry_rank == NULL: station order, = evc_set_station_state), so a checkpoint replays bit for bit. */
int evc_get_entry_rank(evc_engine* e, int16_t* entry_rank);
int evc_set_station_state_ranked(evc_engine* e, const double* remaining_kwh, const int16_t* departure,
                                 const int16_t* est_departure, const int16_t* entry_rank);
int evc_set_env_scalars(evc_engine* e, const int32_t* in_host);
int evc_get_breakdown(evc_engine* e, double* out_host /* [N][3] */);
int evc_set_breakdown(evc_engine* e, const double* in_host);
int evc_clear_status(evc_engine* e);

/* Device-side reduction of the metrics the multi-GPU all-gather carries:
 * out_host[8] = sum profit, sum carbon_cost, sum excess_charge (of running episodes),
 * env-steps executed since create, episodes finished since create, envs with non-zero
 * status; [6] = values a projection solver moved (and tie-snapped to the 2^-16 A grid)
 * since create, [7] = those of them that lay within 1e-6 A of a rounding boundary of env.py:373-378
 * before the snap (the reach of the snap: only there could another solver's rounding differ).  Counted by the
 * iterative slow path always and by the in-row water-filling only on steps with pilots / rates / projected
 * outputs requested (the counting costs the lean streaming kernel 1 us per step). */
int evc_read_metrics(evc_engine* e, double* out_host);

/* Grid of the tie snap: values a projection solver moved are rounded to the nearest point of a
 * 2^-log2_steps_per_amp A grid (offset by sqrt(2)-1 steps) before env.py:373-378's rounding rule.  Default 16.
 * 8 <= log2_steps_per_amp <= 44; at 40 the grid is 2^-40 A ~ 1e-12 A, i.e. the solver's own output to its last
 * bits — how tests/test_gpu_kkt_certificate.py reads the un-snapped optimum of env.py:178-198 out of the product
 * kernels.  Takes effect from the next step; stream-ordered like every other call. */
int evc_set_tie_grid(evc_engine* e, int32_t log2_steps_per_amp);

/* Number of environments the most recent evc_step handed to the iterative/exact projection kernel
 * (diagnostic; synchronises the stream). */
int evc_last_slow_count(evc_engine* e, int32_t* count);

/* Duration (ms) of the most recent evc_step's kernels: with evc_enable_timing(e,1) the streaming kernel
 * and the slow kernel are launched with their own start / stop HIP events on the engine's stream
 * (hipExtLaunchKernel), so the figures are the kernels' begin-to-end times, as a kernel trace reports
 * them; evc_last_step_ms waits for the step.  Used by bench.py's roofline leg.  ms_slow = 0 when the
 * step launched no slow kernel (projection off). */
int evc_enable_timing(evc_engine* e, int32_t on);
int evc_last_step_ms(evc_engine* e, float* ms_main, float* ms_slow);
/* A pipelined step (evc_set_pipeline) is two launches: ms_main above is then the time from the first begin to the last end,
 * and evc_last_half_ms gives each launch's own begin-to-end time (EVC_ESTATE if the last timed step was not pipelined). */
int evc_last_half_ms(evc_engine* e, float* ms_first, float* ms_second);

#ifdef __cplusplus
}
#endif
#endif /* EVCHARGE_H */
