/*
 * battery_dispatch.h — C-ABI of the batched battery-dispatch step (BASELINE config 4,
 * "ElectricityMarketEnv battery-dispatch step").
 *
 * STATUS: a SYNTHETIC WORKLOAD, "parity unpinned".  The reference snapshot contains no
 * implementation of ElectricityMarketEnv — only prose (docs/electricitymarketenv.md:3-27), a
 * commented-out registration (sustaingym/__init__.py:9-12) and a NotImplementedError stub
 * (sustaingym/envs/battery_storage.py:12-37); its market clearing is a multi-period SCED LP over a
 * 24-bus network that is not in the tree.  What is built here is the part of that step that is a
 * data-parallel hot path — the battery's state-of-charge integration, the dispatch clamps, the reward
 * and the observation — with the market reduced to a PRICE-TAKER rule on supplied price traces
 * (SURVEY.md §8f row 3).  Symbols follow the prose: t, e (MWh), a = (a^c, a^d) bids ($/MWh), x
 * dispatch (MWh, > 0 = sold / discharged), p price, l load, m MOER, k forecast steps, T = 288,
 * tau = 5/60 h, reward r = p x + P_CO2 m x - c_T.
 *
 *   dispatch   x = +min(P_max tau, eta_d e)            if p_t >= a^d_0 and not (p_t <= a^c_0)
 *              x = -min(P_max tau, (E_max - e)/eta_c)  if p_t <= a^c_0 and not (p_t >= a^d_0)
 *              x = 0                                   otherwise
 *   energy     e' = e - x/eta_d (x > 0),  e' = e - eta_c x (x < 0)
 *   reward     r = p_t x + P_CO2 m_t x - [t+1 == T] * terminal_price * max(0, e_0 - e')
 *   obs (float32, 4k+6): t+1, e', a_t[2k], x, p_t, l_t, lhat[t+1 .. t+k], m_t, mhat[t+1 .. t+k]
 */
#ifndef BATTERY_DISPATCH_H
#define BATTERY_DISPATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BAT_EPISODE_STEPS 288
#define BAT_TRACE_LEN 289          /* price / load / moer: t = 0..288 */
#define BAT_MAX_FORECAST 36

typedef struct bat_engine bat_engine;

typedef struct bat_config {
    int32_t num_envs;          /* environments on this GPU */
    int32_t forecast_steps;    /* k <= BAT_MAX_FORECAST */
    int32_t bank_slots;        /* resident episodes (traces) */
    int32_t device;
    double capacity_mwh;       /* E_max, 80 in the prose */
    double max_power_mw;       /* P_max */
    double eta_charge, eta_discharge;
    double init_energy_mwh;    /* e_0 */
    double co2_price_per_kg;   /* P_CO2 with MOER in kg CO2 / MWh */
} bat_config;

int bat_create(const bat_config* cfg, bat_engine** out);
void bat_destroy(bat_engine* e);
const char* bat_last_error(void);
int bat_obs_dim(const bat_engine* e);           /* 4k + 6 */
int bat_set_stream(bat_engine* e, void* hip_stream);

/* Episode traces (host): price/load/moer float32 [count][289]; load_fc/moer_fc float32
 * [count][289 + k] (forecast series: lhat[t+1..t+k] = load_fc[t+1..t+k]); terminal_price [count]. */
int bat_upload_traces(bat_engine* e, int32_t first_slot, int32_t count, const float* price,
                      const float* load, const float* load_fc, const float* moer, const float* moer_fc,
                      const double* terminal_price);

/* slots[N] (host, NULL = env i plays slot i % bank_slots).  obs: device float32 [N][4k+6]. */
int bat_reset(bat_engine* e, const int32_t* slots, float* obs_dev);
/* bids: device float32 [N][2k] = (a^c[k], a^d[k]); reward float64 [N]; terminated uint8 [N]. */
int bat_step(bat_engine* e, const float* bids_dev, float* obs_dev, double* reward_dev,
             uint8_t* terminated_dev);
/* T steps in ONE launch (round 4; the f2 pattern of the EV engine): equivalent to `steps` calls of bat_step with the bids of
 * step i = bids_ring_dev[i % ring_len] (device float32 [ring_len][N][2k]); an environment's t, e and running return stay in
 * registers between its steps.  obs / reward / terminated receive the LAST step's outputs.  With obs_traj_dev / reward_traj_dev
 * (device float32 [steps][N][4k+6] / float64 [steps][N]; either may be NULL) EVERY step's observation and reward are also written
 * — the rollout buffer a learner reads: then the launch moves what `steps` calls move (2k bids in, 4k+6 floats out per
 * environment-step) without `steps` launch boundaries.  Steps after an episode's end are no-ops (reward 0 in reward_traj, the
 * observation rows of those steps are not written). */
int bat_rollout(bat_engine* e, const float* bids_ring_dev, int32_t ring_len, int32_t steps, float* obs_dev, double* reward_dev,
                uint8_t* terminated_dev, float* obs_traj_dev, double* reward_traj_dev);
/* The same with the trajectory rows |traj_pitch| floats apart (even, >= 4k+6): obs_traj_dev is [steps][N][|traj_pitch|], row
 * (i, env) starts at ((i * N) + env) * |traj_pitch| and holds 4k+6 floats.
 *   traj_pitch > 0: what lies behind the 4k+6 floats of a row is NOT written (the caller may keep other columns there).
 *   traj_pitch < 0: the floats behind each row, up to the pitch, belong to the kernel and are filled with ZEROS: with a pitch
 *     that is a multiple of 32 floats and a 128-byte aligned base every row leaves as whole 128-byte lines (non-temporal
 *     stores) instead of 600-byte rows that straddle lines (round 5).  -160 (640 B) is what BatteryDispatchVectorEnv.rollout
 *     passes for the buffer it allocates itself; it hands out the [steps, N, 4k+6] view.
 * The bid ring, one step's trajectory slab and the reward trajectory are addressed with 32-bit offsets: each must stay below
 * 4 GiB (ring_len * N * 2k * 4, N * |traj_pitch| * 4, steps * N * 8 bytes) or the call fails. */
int bat_rollout_pitched(bat_engine* e, const float* bids_ring_dev, int32_t ring_len, int32_t steps, float* obs_dev, double* reward_dev,
                        uint8_t* terminated_dev, float* obs_traj_dev, int32_t traj_pitch, double* reward_traj_dev);
/* the same with host buffers (staged through engine-owned device buffers) */
int bat_reset_host(bat_engine* e, const int32_t* slots, float* obs_host);
int bat_step_host(bat_engine* e, const float* bids_host, float* obs_host, double* reward_host,
                  uint8_t* terminated_host);
/* state for inspection / tests: energy [N] float64, t [N] int32 */
int bat_get_state(bat_engine* e, double* energy_host, int32_t* t_host);
/* sums over the shard for the multi-GPU metrics all-gather: out[4] = sum energy, sum of rewards
 * since reset, env-steps since create, environments terminated */
int bat_read_metrics(bat_engine* e, double* out_host);

#ifdef __cplusplus
}
#endif
#endif /* BATTERY_DISPATCH_H */
