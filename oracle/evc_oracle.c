/*
 * evc_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See evc_oracle.h.
 *
 * Scalar restatement of the reference hot path.  Citations: "env.py:L" etc. are relative to
 * /root/reference/sustaingym/envs/evcharging/; "acnportal ..." names the function of the
 * un-vendored dependency acnportal 0.3.x (pyproject.toml:28 `acnportal>=0.3.3`) whose
 * published algorithm is restated here (parity unpinned — see the header).
 *
 * Deliberately object-shaped (EV / battery / EVSE / event-queue structs, one environment at
 * a time) so that it shares no structure with the wave-per-environment HIP kernels.
 */
#include "evc_oracle.h"
#include "evc_oracle_priv.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- constants, env.py:99-114 (same expression order as the Python source) ---- */
static const double TIMESTEP_DURATION = 5.0;
static const double ACTION_SCALE_FACTOR = 32.0;
static const double VOLTAGE = 208.0;
#define MARGINAL_REVENUE_PER_KWH 0.15
#define OPERATING_MARGIN 0.20
#define CO2_COST_PER_METRIC_TON 30.85
#define VIOLATION_WEIGHT 0.001
static double A_MINS_TO_KWH, A_PERS_TO_KWH, PROFIT_FACTOR, VIOLATION_FACTOR, CARBON_COST_FACTOR,
    MARGINAL_PROFIT_PER_KWH;
static int g_consts_ready = 0;
static void init_consts(void) {
    if (g_consts_ready) return;
    MARGINAL_PROFIT_PER_KWH = MARGINAL_REVENUE_PER_KWH * OPERATING_MARGIN;   /* env.py:106 */
    A_MINS_TO_KWH = (1.0 / 60.0) * (VOLTAGE / 1000.0);                       /* env.py:108 */
    A_PERS_TO_KWH = A_MINS_TO_KWH * TIMESTEP_DURATION;                       /* env.py:111 */
    PROFIT_FACTOR = A_PERS_TO_KWH * MARGINAL_PROFIT_PER_KWH;                 /* env.py:112 */
    VIOLATION_FACTOR = A_PERS_TO_KWH * VIOLATION_WEIGHT;                     /* env.py:113 */
    CARBON_COST_FACTOR = A_PERS_TO_KWH * (CO2_COST_PER_METRIC_TON / 1000.0); /* env.py:114 */
    g_consts_ready = 1;
}

/* event_generation.py:60-62 */
static const double BATTERY_CAPACITY = 100.0;
static const double BATTERY_MAX_POWER = 100.0;
/* acnportal Linear2StageBattery default transition_soc */
static const double TRANSITION_SOC = 0.8;
/* acnportal EV.fully_charged: `not (remaining_demand > 1e-3)` */
static const double FULLY_CHARGED_EPS = 1e-3;
#define MAX_TIMESTEP 288 /* env.py:124 */

/* acnportal event precedences: UnplugEvent 0 < PluginEvent 10 < RecomputeEvent 20 */
enum { EV_UNPLUG = 0, EV_PLUGIN = 10, EV_RECOMPUTE = 20 };


typedef struct { /* acns.EV + acns.Linear2StageBattery, event_generation.py:173-186 */
    int arrival, departure, est_departure, station;
    double requested_energy;
    double energy_delivered;
    double capacity, current_charge, max_power;
    double current_charging_rate; /* A */
} ev_t;

typedef struct {
    int ts;
    int precedence;
    int ev; /* index into evs, -1 for recompute */
} event_t;

struct orc_env {
    const orc_net* net;
    int k, project;
    int battery_model;             /* ORC_BATTERY_* (Linear2StageBattery.charge_calculation) */
    ev_t evs[ORC_MAX_SESSIONS];
    int n_evs;
    event_t queue[2 * ORC_MAX_SESSIONS + MAX_TIMESTEP + 8];
    int qlen;
    int evse_ev[ORC_MAX_STATIONS]; /* EVSE._ev: index of plugged EV or -1 */
    int iteration;                 /* Simulator._iteration */
    int t;                         /* EVChargingEnv.t */
    double moer[ORC_MOER_ROWS * ORC_MOER_COLS];
    double breakdown[3];
    float demands[ORC_MAX_STATIONS];        /* env._demands (float32, env.py:138) */
    float est_departures[ORC_MAX_STATIONS]; /* env._est_departures */
    uint32_t status;
    int done;
};

/* ------------------------------------------------------------------------------------ */

orc_net* orc_net_create(int n, int m, const double* A, const double* phase_deg,
                        const double* magnitudes, const uint8_t* evse_kind) {
    init_consts();
    if (n <= 0 || n > ORC_MAX_STATIONS || m < 0 || m > ORC_MAX_CONSTRAINTS) return NULL;
    orc_net* net = (orc_net*)calloc(1, sizeof(orc_net));
    net->n = n;
    net->m = m;
    memcpy(net->A, A, sizeof(double) * (size_t)m * n);
    for (int i = 0; i < n; i++) {
        net->phase_deg[i] = phase_deg[i];
        double rad = phase_deg[i] * (M_PI / 180.0); /* np.deg2rad */
        net->cosphi[i] = cos(rad);
        net->sinphi[i] = sin(rad);
        net->kind[i] = evse_kind[i];
    }
    for (int c = 0; c < m; c++) net->mag[c] = magnitudes[c];
    return net;
}
void orc_net_destroy(orc_net* net) { free(net); }

orc_env* orc_env_create(const orc_net* net, int k, int project) {
    init_consts();
    if (!net || k < 1 || k > 36) return NULL; /* env.py:120 */
    orc_env* e = (orc_env*)calloc(1, sizeof(orc_env));
    e->net = net;
    e->k = k;
    e->project = project;
    e->done = 1; /* must reset first */
    for (int i = 0; i < ORC_MAX_STATIONS; i++) e->evse_ev[i] = -1;
    return e;
}
void orc_env_destroy(orc_env* e) { free(e); }
void orc_env_set_battery_model(orc_env* e, int model) { e->battery_model = model; }
int orc_env_t(const orc_env* e) { return e->t; }

/* acnportal EventQueue: heap of (timestamp, event); Event.__lt__ compares precedence.
 * Restated as a sorted array (ties keep insertion order; ties only occur between events on
 * different stations, for which the order is immaterial). */
static void queue_push(orc_env* e, int ts, int precedence, int ev) {
    int pos = e->qlen;
    while (pos > 0) {
        event_t* p = &e->queue[pos - 1];
        if (p->ts < ts || (p->ts == ts && p->precedence <= precedence)) break;
        e->queue[pos] = *p;
        pos--;
    }
    e->queue[pos].ts = ts;
    e->queue[pos].precedence = precedence;
    e->queue[pos].ev = ev;
    e->qlen++;
}

/* numpy pairwise summation (np.sum over a float64 vector, n <= 128) */
static double np_sum(const double* a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* env.py:381-394 _get_observation + acnportal Interface.active_sessions
 * (active = plugged-in EVs with `not fully_charged`). */
static void get_observation(orc_env* e, float* obs) {
    const int n = e->net->n, k = e->k;
    for (int i = 0; i < n; i++) {
        e->est_departures[i] = 0.0f; /* env.py:383-384 */
        e->demands[i] = 0.0f;
    }
    for (int i = 0; i < n; i++) {
        int idx = e->evse_ev[i];
        if (idx < 0) continue;
        const ev_t* ev = &e->evs[idx];
        double remaining_demand = ev->requested_energy - ev->energy_delivered;
        if (!(remaining_demand > FULLY_CHARGED_EPS)) continue; /* EV.fully_charged */
        e->est_departures[i] = (float)(ev->est_departure - e->t); /* env.py:387 */
        e->demands[i] = (float)remaining_demand;                  /* env.py:388 */
    }
    if (!obs) return;
    /* flattened key order: demands, est_departures, forecasted_moer, prev_moer, timestep */
    for (int i = 0; i < n; i++) obs[i] = e->demands[i];
    for (int i = 0; i < n; i++) obs[n + i] = e->est_departures[i];
    const double* row = &e->moer[(size_t)e->t * ORC_MOER_COLS];
    for (int j = 0; j < k; j++) obs[2 * n + j] = (float)row[1 + j]; /* env.py:391 */
    obs[2 * n + k] = (float)row[0];                                 /* env.py:390 */
    obs[2 * n + k + 1] = (float)((double)e->t / (double)MAX_TIMESTEP); /* env.py:392 */
}

void orc_env_reset(orc_env* e, int n_sessions, const orc_session* s, const double* requested,
                   const double* moer, float* obs_out) {
    const int n = e->net->n;
    if (n_sessions > ORC_MAX_SESSIONS) n_sessions = ORC_MAX_SESSIONS;
    e->n_evs = n_sessions;
    e->qlen = 0;
    for (int i = 0; i < n; i++) e->evse_ev[i] = -1; /* env.py:320 fresh network */
    unsigned char has_arrival[MAX_TIMESTEP + 2];
    memset(has_arrival, 0, sizeof(has_arrival));
    for (int i = 0; i < n_sessions; i++) {
        ev_t* ev = &e->evs[i];
        ev->arrival = s[i].arrival;
        ev->departure = s[i].departure;
        ev->est_departure = s[i].est_departure;
        ev->station = s[i].station;
        ev->requested_energy = requested[i]; /* already capped, event_generation.py:169-170 */
        ev->energy_delivered = 0.0;
        ev->capacity = BATTERY_CAPACITY;
        double init = BATTERY_CAPACITY - requested[i]; /* event_generation.py:175 */
        ev->current_charge = init > 0.0 ? init : 0.0;
        ev->max_power = BATTERY_MAX_POWER;
        ev->current_charging_rate = 0.0;
        queue_push(e, ev->arrival, EV_PLUGIN, i); /* event_generation.py:189-190 */
        if (ev->arrival >= 0 && ev->arrival <= MAX_TIMESTEP) has_arrival[ev->arrival] = 1;
    }
    for (int ts = 0; ts <= MAX_TIMESTEP; ts++) /* event_generation.py:199-203 */
        if (!has_arrival[ts]) queue_push(e, ts, EV_RECOMPUTE, -1);
    memcpy(e->moer, moer, sizeof(double) * ORC_MOER_ROWS * ORC_MOER_COLS); /* env.py:323 */
    e->iteration = 0;
    e->t = 0; /* env.py:329 */
    e->breakdown[0] = e->breakdown[1] = e->breakdown[2] = 0.0; /* env.py:332-333 */
    e->status = 0;
    e->done = 0;
    get_observation(e, obs_out); /* env.py:338 */
}

/* acnportal Linear2StageBattery._charge_stepwise (charge_calculation="stepwise", noise 0): the
 * legacy model, "should only be used for reproducing results from older versions of acnportal".
 * Returns the charging power in kW. */
static double battery_charge_stepwise(ev_t* ev, double pilot, double voltage, double period) {
    double rate_to_full = (ev->capacity - ev->current_charge) / (period / 60.0);
    double soc = ev->current_charge / ev->capacity;
    double pilot_kw = pilot * voltage / 1000.0;
    double charge_power;
    if (soc < TRANSITION_SOC) {
        charge_power = fmin(fmin(pilot_kw, ev->max_power), rate_to_full);
    } else {
        double taper = (1.0 - soc) / (1.0 - TRANSITION_SOC) * ev->max_power;
        charge_power = fmin(fmin(pilot_kw, taper), rate_to_full);
    }
    ev->current_charge += charge_power * (period / 60.0);
    return charge_power;
}

/* acnportal Linear2StageBattery._charge (charge_calculation="continuous", the constructor's DEFAULT
 * in acnportal >= 0.3; the reference builds the battery without that argument,
 * event_generation.py:173-176), noise 0.  All arithmetic in state of charge, in acnportal's
 * operation order: the maximum rate is constant up to the transition SoC and then falls linearly to 0
 * at SoC 1, integrated exactly over the period (hence the exponentials).  Returns kW. */
static double battery_charge_continuous(ev_t* ev, double pilot, double voltage, double period) {
    double soc = ev->current_charge / ev->capacity;
    /* pilot and maximum rate as changes of SoC per period */
    double pilot_dsoc = pilot * voltage / 1000.0 / ev->capacity / (60.0 / period);
    double max_dsoc = ev->max_power / ev->capacity / (60.0 / period);
    if (pilot_dsoc > max_dsoc) pilot_dsoc = max_dsoc;
    /* SoC at which a battery charging at pilot_dsoc meets the falling maximum-rate line */
    double pilot_transition_soc =
        TRANSITION_SOC + (pilot_dsoc - max_dsoc) / max_dsoc * (TRANSITION_SOC - 1.0);
    double curr_soc;
    if (soc < pilot_transition_soc) {
        if (1.0 <= (pilot_transition_soc - soc) / pilot_dsoc) {
            curr_soc = pilot_dsoc + soc; /* stays in the constant-rate region */
        } else {                          /* crosses into the ramp-down region within the period */
            curr_soc = 1.0 + exp((pilot_dsoc + soc - pilot_transition_soc) / (pilot_transition_soc - 1.0)) *
                                 (pilot_transition_soc - 1.0);
        }
    } else {
        curr_soc = 1.0 + exp(pilot_dsoc / (pilot_transition_soc - 1.0)) * (soc - 1.0);
    }
    double dsoc = curr_soc - soc;
    ev->current_charge = curr_soc * ev->capacity;
    return dsoc * ev->capacity / (period / 60.0); /* average power over the period */
}

/* Linear2StageBattery.charge followed by acnportal EV.charge; returns the actual charging rate (A). */
static double ev_charge(ev_t* ev, double pilot, double voltage, double period, int battery_model) {
    if (pilot == 0.0) {
        ev->current_charging_rate = 0.0;
        return 0.0;
    }
    double charge_power = battery_model == ORC_BATTERY_STEPWISE
                              ? battery_charge_stepwise(ev, pilot, voltage, period)
                              : battery_charge_continuous(ev, pilot, voltage, period);
    double charge_rate = charge_power * 1000.0 / voltage;
    /* EV.charge */
    ev->energy_delivered += (charge_rate * voltage / 1000.0) * (period / 60.0);
    ev->current_charging_rate = charge_rate;
    return charge_rate;
}

int orc_project_action_impl(const orc_net* net, const double* action, const float* demands,
                            double* x_out, double* kkt_out); /* evc_oracle_proj.c */

int orc_project_action(const orc_net* net, const double* action, const float* demands,
                       double* x_out, double* kkt_out) {
    init_consts();
    return orc_project_action_impl(net, action, demands, x_out, kkt_out);
}

/* env.py:340-379 _to_schedule; action is the (already clamped) normalised action, float64 */
static void to_schedule(orc_env* e, const double* action_in, double* pilots, double* projected) {
    const orc_net* net = e->net;
    const int n = net->n;
    double action[ORC_MAX_STATIONS];
    if (e->project) { /* env.py:363-364 */
        double kkt[4];
        int rc = orc_project_action_impl(net, action_in, e->demands, action, kkt);
        if (rc != 0) e->status |= ORC_STATUS_PROJ_NOCONV;
    } else {
        for (int i = 0; i < n; i++) action[i] = action_in[i];
    }
    for (int i = 0; i < n; i++) projected[i] = action[i];
    for (int i = 0; i < n; i++) {
        double a = action[i] * ACTION_SCALE_FACTOR; /* env.py:366 */
        if (net->kind[i] == 0) {                    /* min pilot 6, env.py:373-375 */
            pilots[i] = (a >= 6.0) ? rint(a) : 0.0; /* np.round = half-to-even */
        } else {
            pilots[i] = rint(a / 8.0) * 8.0; /* env.py:378 */
        }
    }
}

/* acnportal Simulator.step(new_schedule) for one loop pass (every timestep has an event,
 * event_generation.py:199-203, and env.py:284 clears _resolve). */
static int simulator_step(orc_env* e, const double* pilots, double* rates) {
    const int n = e->net->n;
    if (e->qlen == 0) return 1;
    /* network.update_pilots(pilot_signals, _iteration, period): EVSE.set_pilot -> EV.charge */
    for (int i = 0; i < n; i++) {
        int idx = e->evse_ev[i];
        rates[i] = (idx >= 0) ? ev_charge(&e->evs[idx], pilots[i], VOLTAGE, TIMESTEP_DURATION, e->battery_model) : 0.0;
    }
    e->iteration += 1;
    /* event_queue.get_current_events(_iteration): pop everything with timestamp <= it FIRST,
     * then process; events pushed while processing are seen by the next pass. */
    int ncur = 0;
    while (ncur < e->qlen && e->queue[ncur].ts <= e->iteration) ncur++;
    event_t cur[2 * ORC_MAX_SESSIONS + 8];
    int ncopy = ncur;
    if (ncopy > (int)(sizeof(cur) / sizeof(cur[0]))) ncopy = (int)(sizeof(cur) / sizeof(cur[0]));
    memcpy(cur, e->queue, sizeof(event_t) * (size_t)ncopy);
    memmove(e->queue, e->queue + ncur, sizeof(event_t) * (size_t)(e->qlen - ncur));
    e->qlen -= ncur;
    for (int j = 0; j < ncopy; j++) { /* Simulator._process_event */
        const event_t* evn = &cur[j];
        if (evn->precedence == EV_PLUGIN) {
            ev_t* ev = &e->evs[evn->ev];
            if (e->evse_ev[ev->station] >= 0) {
                /* acnportal raises StationOccupiedError; we flag and skip the session */
                e->status |= ORC_STATUS_OCCUPIED;
                continue;
            }
            e->evse_ev[ev->station] = evn->ev;
            queue_push(e, ev->departure, EV_UNPLUG, evn->ev);
        } else if (evn->precedence == EV_UNPLUG) {
            ev_t* ev = &e->evs[evn->ev];
            if (e->evse_ev[ev->station] == evn->ev) e->evse_ev[ev->station] = -1;
        }
    }
    return e->qlen == 0;
}

/* env.py:431-464 _get_reward */
static double get_reward(orc_env* e, const double* pilots, const double* rates) {
    const orc_net* net = e->net;
    const int n = net->n, m = net->m;
    double total_charging_rate = np_sum(rates, n); /* env.py:445 */
    double profit = PROFIT_FACTOR * total_charging_rate;
    /* acnportal ChargingNetwork.constraint_current: constraint_matrix @ (schedule * exp(j phi)) */
    double excess_terms[ORC_MAX_CONSTRAINTS];
    for (int c = 0; c < m; c++) {
        double re = 0.0, im = 0.0;
        for (int i = 0; i < n; i++) {
            double a = net->A[c * n + i];
            re += a * (pilots[i] * net->cosphi[i]);
            im += a * (pilots[i] * net->sinphi[i]);
        }
        double current_sum = hypot(re, im); /* np.abs(complex) */
        double ex = current_sum - net->mag[c];
        excess_terms[c] = ex > 0.0 ? ex : 0.0; /* env.py:451 */
    }
    double excess_current = np_sum(excess_terms, m);
    double excess_charge = excess_current * VIOLATION_FACTOR; /* env.py:452 */
    double carbon_cost = CARBON_COST_FACTOR * total_charging_rate *
                         e->moer[(size_t)e->t * ORC_MOER_COLS + 0]; /* env.py:455 */
    double total_reward = profit - carbon_cost - excess_charge;
    e->breakdown[0] += profit;
    e->breakdown[1] += carbon_cost;
    e->breakdown[2] += excess_charge;
    return total_reward;
}

static void step_f64(orc_env* e, const double* action, float* obs_out, orc_step_result* res) {
    const int n = e->net->n;
    memset(res, 0, sizeof(*res));
    if (e->done) { /* the reference would fail inside acnportal; we flag and ignore */
        e->status |= ORC_STATUS_STEP_AFTER_DONE;
        res->status = e->status;
        res->terminated = 1;
        memcpy(res->breakdown, e->breakdown, sizeof(res->breakdown));
        get_observation(e, obs_out);
        return;
    }
    e->t += 1; /* env.py:279 */
    double clamped[ORC_MAX_STATIONS];
    for (int i = 0; i < n; i++) {
        double a = action[i];
        if (!(a >= 0.0)) { /* also catches NaN */
            if (a != 0.0) e->status |= ORC_STATUS_ACTION_CLAMPED;
            a = 0.0;
        } else if (a > 1.0) {
            e->status |= ORC_STATUS_ACTION_CLAMPED;
            a = 1.0;
        }
        clamped[i] = a;
    }
    to_schedule(e, clamped, res->pilots, res->projected); /* env.py:282 */
    int done = simulator_step(e, res->pilots, res->rates); /* env.py:283 */
    get_observation(e, obs_out);                            /* env.py:287 */
    res->reward = get_reward(e, res->pilots, res->rates);   /* env.py:288 */
    res->terminated = done;
    e->done = done;
    memcpy(res->breakdown, e->breakdown, sizeof(res->breakdown));
    res->status = e->status;
}

void orc_env_step(orc_env* e, const float* action, float* obs_out, orc_step_result* res) {
    double a[ORC_MAX_STATIONS];
    for (int i = 0; i < e->net->n; i++) a[i] = (double)action[i];
    step_f64(e, a, obs_out, res);
}

void orc_env_step_discrete(orc_env* e, const int64_t* action, int bins, float* obs_out,
                           orc_step_result* res) {
    /* wrappers.py:43-45: np.asarray(action, dtype=float32) / (bins - 1)  (float32 division) */
    double a[ORC_MAX_STATIONS];
    for (int i = 0; i < e->net->n; i++) {
        float f = (float)action[i] / (float)(bins - 1);
        a[i] = (double)f;
    }
    step_f64(e, a, obs_out, res);
}

void orc_env_station_state(const orc_env* e, double* remaining, int16_t* departure,
                           int16_t* est_departure) {
    for (int i = 0; i < e->net->n; i++) {
        int idx = e->evse_ev[i];
        if (idx < 0) {
            remaining[i] = 0.0;
            departure[i] = -1;
            est_departure[i] = 0;
        } else {
            const ev_t* ev = &e->evs[idx];
            remaining[i] = ev->requested_energy - ev->energy_delivered;
            departure[i] = (int16_t)ev->departure;
            est_departure[i] = (int16_t)ev->est_departure;
        }
    }
}

/* env.py:422-429 */
double orc_max_profit(int n_sessions, const orc_session* s, const double* requested) {
    init_consts();
    double terms[ORC_MAX_SESSIONS];
    if (n_sessions > ORC_MAX_SESSIONS) n_sessions = ORC_MAX_SESSIONS;
    for (int i = 0; i < n_sessions; i++) {
        double duration = (double)(s[i].departure - s[i].arrival);
        double max_kwh = duration * ACTION_SCALE_FACTOR * A_PERS_TO_KWH;
        double provide = requested[i] < max_kwh ? requested[i] : max_kwh;
        terms[i] = provide * MARGINAL_PROFIT_PER_KWH;
    }
    /* np.sum with pairwise blocks (n <= 128 path; longer inputs split recursively) */
    if (n_sessions <= 128) return np_sum(terms, n_sessions);
    int n2 = n_sessions / 2;
    n2 -= n2 % 8;
    return np_sum(terms, n2) + np_sum(terms + n2, n_sessions - n2);
}

/* ------------------------------- batch driver ---------------------------------------- */

struct orc_batch {
    const orc_net* net;
    int N, k, project;
    orc_env** envs;
    orc_env* block;
    int32_t* slot;
    /* bank (borrowed copies) */
    int bank_slots, stride, moer_days;
    int32_t* n_sessions;
    orc_session* sessions;
    double* requested;
    int32_t* moer_day;
    double* moer;
};

orc_batch* orc_batch_create(const orc_net* net, int N, int k, int project) {
    orc_batch* b = (orc_batch*)calloc(1, sizeof(orc_batch));
    b->net = net;
    b->N = N;
    b->k = k;
    b->project = project;
    b->envs = (orc_env**)calloc((size_t)N, sizeof(orc_env*));
    b->slot = (int32_t*)calloc((size_t)N, sizeof(int32_t));
    /* ONE block for all environments (an orc_env is ~112 kB: as 65 536 separate callocs a full-size batch grew
     * the brk heap to 7.8 GB and shrank it again at destroy; the GPU test-suite's intermittent "Memory access
     * fault by GPU" hit exactly such addresses shortly afterwards).  A block this size is
     * mmap'ed and unmapped as a whole. */
    b->block = (orc_env*)calloc((size_t)N, sizeof(orc_env));
    for (int i = 0; i < N; i++) {
        orc_env* e = &b->block[i];
        e->net = net;
        e->k = k;
        e->project = project;
        e->done = 1;
        for (int j = 0; j < ORC_MAX_STATIONS; j++) e->evse_ev[j] = -1;
        b->envs[i] = e;
    }
    return b;
}

static void batch_free_bank(orc_batch* b) {
    free(b->n_sessions);
    free(b->sessions);
    free(b->requested);
    free(b->moer_day);
    free(b->moer);
    b->n_sessions = NULL;
    b->sessions = NULL;
    b->requested = NULL;
    b->moer_day = NULL;
    b->moer = NULL;
}

void orc_batch_destroy(orc_batch* b) {
    if (!b) return;
    free(b->block);
    free(b->envs);
    free(b->slot);
    batch_free_bank(b);
    free(b);
}

void orc_batch_set_battery_model(orc_batch* b, int model) {
    for (int i = 0; i < b->N; i++) orc_env_set_battery_model(b->envs[i], model);
}

void orc_batch_set_bank(orc_batch* b, int bank_slots, int stride, const int32_t* n_sessions,
                        const orc_session* sessions, const double* requested,
                        const int32_t* moer_day, int moer_days, const double* moer) {
    batch_free_bank(b);
    b->bank_slots = bank_slots;
    b->stride = stride;
    b->moer_days = moer_days;
    size_t ns = (size_t)bank_slots * stride;
    b->n_sessions = (int32_t*)malloc(sizeof(int32_t) * bank_slots);
    b->sessions = (orc_session*)malloc(sizeof(orc_session) * ns);
    b->requested = (double*)malloc(sizeof(double) * ns);
    b->moer_day = (int32_t*)malloc(sizeof(int32_t) * bank_slots);
    size_t msz = (size_t)moer_days * ORC_MOER_ROWS * ORC_MOER_COLS;
    b->moer = (double*)malloc(sizeof(double) * msz);
    memcpy(b->n_sessions, n_sessions, sizeof(int32_t) * bank_slots);
    memcpy(b->sessions, sessions, sizeof(orc_session) * ns);
    memcpy(b->requested, requested, sizeof(double) * ns);
    memcpy(b->moer_day, moer_day, sizeof(int32_t) * bank_slots);
    memcpy(b->moer, moer, sizeof(double) * msz);
}

static void batch_reset_one(orc_batch* b, int i, int slot, float* obs_row) {
    b->slot[i] = slot;
    size_t off = (size_t)slot * b->stride;
    orc_env_reset(b->envs[i], b->n_sessions[slot], b->sessions + off, b->requested + off,
                  b->moer + (size_t)b->moer_day[slot] * ORC_MOER_ROWS * ORC_MOER_COLS, obs_row);
}

void orc_batch_reset(orc_batch* b, const int32_t* slots, float* obs) {
    const int F = 2 * b->net->n + b->k + 2;
    for (int i = 0; i < b->N; i++) {
        int slot = slots ? slots[i] : (i % b->bank_slots);
        batch_reset_one(b, i, slot, obs ? obs + (size_t)i * F : NULL);
    }
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_batch_step(orc_batch* b, const float* actions, const int64_t* discrete, int bins,
                    int autoreset, int autoreset_stride, int threads, float* obs, double* reward,
                    uint8_t* terminated, double* breakdown, float* final_obs, double* pilots,
                    double* rates, double* projected, uint32_t* status) {
    const int n = b->net->n;
    const int F = 2 * n + b->k + 2;
#ifdef _OPENMP
    int nt = threads > 0 ? threads : omp_get_max_threads();
#pragma omp parallel for num_threads(nt) schedule(static)
#else
    (void)threads;
#endif
    for (int i = 0; i < b->N; i++) {
        orc_step_result res;
        float* row = obs + (size_t)i * F;
        if (discrete)
            orc_env_step_discrete(b->envs[i], discrete + (size_t)i * n, bins, row, &res);
        else
            orc_env_step(b->envs[i], actions + (size_t)i * n, row, &res);
        reward[i] = res.reward;
        terminated[i] = (uint8_t)res.terminated;
        if (breakdown) memcpy(breakdown + (size_t)i * 3, res.breakdown, sizeof(double) * 3);
        if (pilots) memcpy(pilots + (size_t)i * n, res.pilots, sizeof(double) * n);
        if (rates) memcpy(rates + (size_t)i * n, res.rates, sizeof(double) * n);
        if (projected) memcpy(projected + (size_t)i * n, res.projected, sizeof(double) * n);
        if (status) status[i] = res.status;
        if (res.terminated && autoreset) {
            if (final_obs) memcpy(final_obs + (size_t)i * F, row, sizeof(float) * F);
            int next = (int)(((long)b->slot[i] + autoreset_stride) % b->bank_slots);
            batch_reset_one(b, i, next, row);
        }
    }
}

void orc_batch_station_state(const orc_batch* b, double* remaining, int16_t* departure,
                             int16_t* est_departure) {
    const int n = b->net->n;
    for (int i = 0; i < b->N; i++)
        orc_env_station_state(b->envs[i], remaining + (size_t)i * n, departure + (size_t)i * n,
                              est_departure + (size_t)i * n);
}

/* The episode loop of BaseAlgorithm.run (sustaingym/algorithms/base.py:63-88) for the two arithmetic-free
 * baselines, every environment of the batch on its own (OpenMP over environments):
 *   policy 2 = GreedyAlgorithm.get_action (algorithms/evcharging/baselines.py:32-35): 1 where the observed
 *              demand is non-zero, else 0;
 *   policy 3 = RandomAlgorithm.get_action (baselines.py:45-51) on the counter-based stream of
 *              orc_random_action (seed, env_id_base + i, episodes done, period t).
 * obs [N][F]: in = the current observations (what reset / the previous step returned), out = those after the
 * last step.  returns[i] += every reward.  Without autoreset a finished environment stops stepping (reward 0,
 * terminated 1), like the reference's `while not done`. */
void orc_batch_rollout(orc_batch* b, int policy, int bins, uint64_t seed, uint32_t env_id_base,
                       int32_t* episodes /* [N] in/out */, int steps, int autoreset, int autoreset_stride,
                       int threads, float* obs, double* reward, uint8_t* terminated, double* breakdown,
                       float* final_obs, double* returns, uint32_t* status) {
    const int n = b->net->n;
    const int F = 2 * n + b->k + 2;
#ifdef _OPENMP
    int nt = threads > 0 ? threads : omp_get_max_threads();
#pragma omp parallel for num_threads(nt) schedule(dynamic, 16)
#else
    (void)threads;
#endif
    for (int i = 0; i < b->N; i++) {
        orc_env* e = b->envs[i];
        float* row = obs + (size_t)i * F;
        float act[ORC_MAX_STATIONS];
        int64_t level[ORC_MAX_STATIONS];
        orc_step_result res;
        uint32_t st = 0;
        for (int s = 0; s < steps; s++) {
            if (e->done) {                       /* step after termination: ignored */
                reward[i] = 0.0;
                terminated[i] = 1;
                break;
            }
            if (policy == 2) {
                for (int j = 0; j < n; j++) act[j] = row[j] > 0.0f ? 1.0f : 0.0f;
            } else {
                orc_random_action(seed, env_id_base + (uint32_t)i, (uint32_t)episodes[i], (uint32_t)e->t, n, bins, act);
            }
            (void)level;
            orc_env_step(e, act, row, &res);
            st |= res.status;
            reward[i] = res.reward;
            terminated[i] = (uint8_t)res.terminated;
            if (returns) returns[i] += res.reward;
            if (breakdown) memcpy(breakdown + (size_t)i * 3, res.breakdown, sizeof(double) * 3);
            if (res.terminated) {
                episodes[i] += 1;
                if (autoreset) {
                    if (final_obs) memcpy(final_obs + (size_t)i * F, row, sizeof(float) * F);
                    int next = (int)(((long)b->slot[i] + autoreset_stride) % b->bank_slots);
                    batch_reset_one(b, i, next, row);
                }
            }
        }
        if (status) status[i] = st;
    }
}

