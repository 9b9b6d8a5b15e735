/*
 * bat_oracle.c — CPU ORACLE (test infrastructure, NOT product code) of the battery-dispatch step.
 *
 * PARITY UNPINNED: the reference has no implementation of ElectricityMarketEnv (only
 * docs/electricitymarketenv.md:3-27); this restates, scalar and one environment at a time, the
 * synthetic step specified in include/battery_dispatch.h / DESIGN.md §10 and pins nothing against the
 * reference.  It exists so that the GPU kernel is checked against an independently written
 * implementation of the same specification (tests/test_gpu_battery.py, tests/test_battery_oracle.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define T_STEPS 288
#define TRACE 289

typedef struct bor_env {
    int k;
    double cap, step_mwh, eta_c, eta_d, e0, pco2;
    double energy;
    int t;
    const float *price, *load, *moer, *load_fc, *moer_fc;   /* this episode's traces */
    double terminal_price;
} bor_env;

bor_env* bor_create(int k, double capacity_mwh, double max_power_mw, double eta_c, double eta_d, double e0,
                    double pco2) {
    bor_env* b = (bor_env*)calloc(1, sizeof(bor_env));
    b->k = k; b->cap = capacity_mwh; b->step_mwh = max_power_mw * (5.0 / 60.0);
    b->eta_c = eta_c; b->eta_d = eta_d; b->e0 = e0; b->pco2 = pco2;
    return b;
}
void bor_destroy(bor_env* b) { free(b); }

static void observe(const bor_env* b, const float* bids, float x, float p, float l, float m, float* obs) {
    const int k = b->k;
    int o = 0;
    obs[o++] = (float)b->t;
    obs[o++] = (float)b->energy;
    for (int i = 0; i < 2 * k; i++) obs[o++] = bids ? bids[i] : 0.0f;
    obs[o++] = x; obs[o++] = p; obs[o++] = l;
    for (int i = 0; i < k; i++) obs[o++] = b->load_fc[b->t + 1 + i];
    obs[o++] = m;
    for (int i = 0; i < k; i++) obs[o++] = b->moer_fc[b->t + 1 + i];
}

void bor_reset(bor_env* b, const float* price, const float* load, const float* load_fc, const float* moer,
               const float* moer_fc, double terminal_price, float* obs) {
    b->price = price; b->load = load; b->load_fc = load_fc; b->moer = moer; b->moer_fc = moer_fc;
    b->terminal_price = terminal_price;
    b->energy = b->e0;
    b->t = 0;
    observe(b, NULL, 0.0f, 0.0f, 0.0f, 0.0f, obs);
}

/* returns terminated */
int bor_step(bor_env* b, const float* bids, float* obs, double* reward) {
    if (b->t >= T_STEPS) { *reward = 0.0; return 1; }
    const double p = (double)b->price[b->t], m = (double)b->moer[b->t];
    const double bid_c = (double)bids[0], bid_d = (double)bids[b->k];
    const int sell = p >= bid_d, buy = p <= bid_c;
    double x = 0.0;
    if (sell && !buy) {                    /* discharge: limited by power and by the energy in store */
        x = b->step_mwh < b->eta_d * b->energy ? b->step_mwh : b->eta_d * b->energy;
        b->energy -= x / b->eta_d;
    } else if (buy && !sell) {             /* charge: limited by power and by the free capacity */
        const double room = (b->cap - b->energy) / b->eta_c;
        x = -(b->step_mwh < room ? b->step_mwh : room);
        b->energy -= b->eta_c * x;
    }
    if (b->energy < 0.0) b->energy = 0.0;
    if (b->energy > b->cap) b->energy = b->cap;
    double r = p * x + b->pco2 * m * x;
    const float pf = b->price[b->t], lf = b->load[b->t], mf = b->moer[b->t];
    b->t += 1;
    const int done = b->t >= T_STEPS;
    if (done) {
        const double short_mwh = b->e0 - b->energy;
        r -= b->terminal_price * (short_mwh > 0.0 ? short_mwh : 0.0);
    }
    observe(b, bids, (float)x, pf, lf, mf, obs);
    *reward = r;
    return done;
}

double bor_energy(const bor_env* b) { return b->energy; }
int bor_t(const bor_env* b) { return b->t; }
