"""Known answers of the battery-dispatch specification (include/battery_dispatch.h) on the oracle
(oracle/bat_oracle.c).  Parity unpinned: the reference has no implementation; these
are hand-derived from the specification's formulas."""
import numpy as np

from oracle.binding import OracleBattery


def _flat_traces(k, price=50.0, load=3000.0, moer=400.0):
    return (np.full(289, price, np.float32), np.full(289, load, np.float32), np.full(289 + k, load + 1, np.float32),
            np.full(289, moer, np.float32), np.full(289 + k, moer + 2, np.float32))


def test_dispatch_rule_and_energy_integration():
    k = 4
    env = OracleBattery(k=k, capacity_mwh=80, max_power_mw=20, eta_charge=0.95, eta_discharge=0.95,
                        init_energy_mwh=40, co2_price_per_kg=0.03)
    price, load, load_fc, moer, moer_fc = _flat_traces(k)
    obs = env.reset(price, load, load_fc, moer, moer_fc, terminal_price=50.0)
    assert obs.shape == (4 * k + 6,) and obs[0] == 0 and obs[1] == 40 and not obs[2:2 + 2 * k].any()
    assert np.all(obs[5 + 2 * k:5 + 3 * k] == 3001) and np.all(obs[6 + 3 * k:] == 402)
    step = 20 * 5 / 60
    # sell: p = 50 >= a^d_0 = 40, and not p <= a^c_0 = 10
    bids = np.array([10, 0, 0, 0, 40, 0, 0, 0], np.float32)
    obs, r, done = env.step(bids)
    assert abs(env.energy - (40 - step / 0.95)) < 1e-12
    assert abs(r - (50 * step + 0.03 * 400 * step)) < 1e-9 and not done
    assert obs[0] == 1 and np.array_equal(obs[2:2 + 2 * k], bids) and abs(obs[2 + 2 * k] - step) < 1e-6
    assert obs[3 + 2 * k] == 50 and obs[4 + 2 * k] == 3000 and obs[5 + 3 * k] == 400
    # buy: p = 50 <= a^c_0 = 60, a^d_0 = 70 not reached
    e_before = env.energy
    obs, r, done = env.step(np.array([60, 0, 0, 0, 70, 0, 0, 0], np.float32))
    assert abs(env.energy - (e_before + 0.95 * step)) < 1e-12 and abs(r + (50 * step + 0.03 * 400 * step)) < 1e-9
    # both conditions hold (a^c >= p >= a^d) -> no dispatch; neither -> no dispatch
    for b in ([60, 0, 0, 0, 40, 0, 0, 0], [10, 0, 0, 0, 70, 0, 0, 0]):
        e_before = env.energy
        obs, r, done = env.step(np.array(b, np.float32))
        assert env.energy == e_before and r == 0.0 and obs[2 + 2 * k] == 0


def test_clamps_and_terminal_cost():
    k = 2
    env = OracleBattery(k=k, capacity_mwh=3.0, max_power_mw=20, eta_charge=0.9, eta_discharge=0.8,
                        init_energy_mwh=1.0, co2_price_per_kg=0.0)
    price, load, load_fc, moer, moer_fc = _flat_traces(k, price=20.0)
    env.reset(price, load, load_fc, moer, moer_fc, terminal_price=30.0)
    sell = np.array([0, 0, 10, 0], np.float32)
    obs, r, done = env.step(sell)                 # only eta_d * e = 0.8 MWh can be sold
    assert abs(r - 20 * 0.8) < 1e-12 and env.energy == 0.0
    obs, r, done = env.step(sell)                 # empty: nothing to sell
    assert r == 0.0 and env.energy == 0.0
    buy = np.array([25, 0, 99, 0], np.float32)
    obs, r, done = env.step(buy)                  # full power: 1.6667 MWh bought, 1.5 stored
    assert abs(env.energy - 0.9 * 20 * 5 / 60) < 1e-12
    obs, r, done = env.step(buy)
    obs, r, done = env.step(buy)                  # capacity reached: buys only the room / eta_c
    assert abs(env.energy - 3.0) < 1e-12
    obs, r, done = env.step(buy)
    assert r == 0.0
    total = 6
    while not done:
        obs, r, done = env.step(sell if total == 287 else np.array([0, 0, 99, 0], np.float32))
        total += 1
    assert total == 288 and obs[0] == 288
    # the last step sold 1.6667 MWh of 3.0 -> e_T = 3 - 1.6667/0.8 = 0.9167 < e_0 = 1: shortfall 0.0833 MWh
    e_T = 3.0 - (20 * 5 / 60) / 0.8
    assert abs(env.energy - e_T) < 1e-12
    assert abs(r - (20 * (20 * 5 / 60) - 30.0 * (1.0 - e_T))) < 1e-9
    obs, r, done = env.step(sell)                 # after termination: no-op
    assert r == 0.0 and done
