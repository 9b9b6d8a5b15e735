"""Synthetic episodes (event tables + MOER days) of the shape the reference generates.

Used by bench.py, smoke() and the parity tests where the packaged traces are not needed
(SURVEY.md §8d, configs 2-3): per episode a session count, arrivals / durations / requested
energies drawn uniformly, and — like ``GMMsTraceGenerator._create_events``
(event_generation.py:489-514) — stations assigned in arrival order among the EVSEs that are
free at the arrival time (a session that finds no free EVSE is dropped).
"""
from __future__ import annotations

import numpy as np

from ._lib import SESSION_DTYPE, MOER_ROWS, MOER_COLS


def synthetic_moer(num_days: int, seed: int = 0) -> np.ndarray:
    """float64 [num_days, 289, 37] in [0,1]: column 0 history, columns 1..36 forecasts."""
    rng = np.random.default_rng(seed)
    t = np.arange(MOER_ROWS + MOER_COLS)[None, :]
    phase = rng.uniform(0, 2 * np.pi, (num_days, 1))
    base = 0.35 + 0.2 * np.sin(2 * np.pi * t / 288.0 + phase) + 0.05 * rng.standard_normal((num_days, MOER_ROWS + MOER_COLS))
    base = np.clip(base, 0.0, 1.0)
    moer = np.empty((num_days, MOER_ROWS, MOER_COLS))
    for j in range(MOER_COLS):
        moer[:, :, j] = base[:, j:j + MOER_ROWS]
    moer[:, :, 1:] = np.clip(moer[:, :, 1:] + 0.01 * rng.standard_normal((num_days, MOER_ROWS, MOER_COLS - 1)), 0.0, 1.0)
    return np.round(moer, 6)


def synthetic_episodes(num_episodes: int, num_stations: int, seed: int = 0,
                       min_sessions: int = 5, max_sessions: int = 40, stride: int | None = None,
                       max_arrival: int = 200, min_duration: int = 6, max_duration: int = 96,
                       min_kwh: float = 2.0, max_kwh: float = 60.0, moer_days: int = 1,
                       early_fraction: float = 0.0):
    """Returns ``(n_sessions[P], sessions[P, stride], requested[P, stride], moer_day[P])``.

    ``early_fraction``: fraction of sessions forced to arrive in periods 0/1 (exercises the
    "arrival 0 plugs at iteration 1" ordering).
    """
    rng = np.random.default_rng(seed)
    P, n = num_episodes, num_stations
    E = max_sessions
    stride = stride or E
    assert stride >= E
    counts = rng.integers(min_sessions, max_sessions + 1, P)
    arrival = rng.integers(0, max_arrival + 1, (P, E))
    if early_fraction > 0:
        early = rng.random((P, E)) < early_fraction
        arrival = np.where(early, rng.integers(0, 2, (P, E)), arrival)
    duration = rng.integers(min_duration, max_duration + 1, (P, E))
    # a few very short sessions (departure == arrival bin happens in real traces)
    duration = np.where(rng.random((P, E)) < 0.02, 0, duration)
    requested = rng.uniform(min_kwh, max_kwh, (P, E))
    valid = np.arange(E)[None, :] < counts[:, None]
    arrival = np.where(valid, arrival, 10_000)
    order = np.argsort(arrival, axis=1, kind='stable')
    arrival = np.take_along_axis(arrival, order, 1)
    duration = np.take_along_axis(duration, order, 1)
    requested = np.take_along_axis(requested, order, 1)
    valid = np.take_along_axis(valid, order, 1)
    departure = np.minimum(arrival + duration, 287)
    est = np.clip(departure + rng.integers(-12, 13, (P, E)), arrival + 1, 287)

    station = np.full((P, E), -1, dtype=np.int64)
    # An EVSE is free for a new plug-in from the simulator pass in which its previous EV is
    # unplugged: the EV plugs at pass max(arrival, 1) and its Unplug event is seen at pass
    # max(departure, plug pass + 1) (acnportal pops events before processing them).  Plugging
    # earlier would raise StationOccupiedError in acnportal.
    free_pass = np.zeros((P, n), dtype=np.int64)
    rows = np.arange(P)
    for j in range(E):
        plug_pass = np.maximum(arrival[:, j], 1)
        avail = free_pass <= plug_pass[:, None]
        score = np.where(avail, rng.random((P, n)), -1.0)
        pick = np.argmax(score, axis=1)
        ok = valid[:, j] & avail[rows, pick]
        station[:, j] = np.where(ok, pick, -1)
        release = np.maximum(departure[:, j], plug_pass + 1)
        free_pass[rows, pick] = np.where(ok, release, free_pass[rows, pick])
    keep = station >= 0
    # compact kept sessions to the front, preserving arrival order
    order = np.argsort(~keep, axis=1, kind='stable')
    n_sessions = keep.sum(axis=1).astype(np.int32)
    sess = np.zeros((P, stride), dtype=SESSION_DTYPE)
    req = np.zeros((P, stride), dtype=np.float64)
    a = np.take_along_axis(arrival, order, 1)
    d = np.take_along_axis(departure, order, 1)
    e = np.take_along_axis(est, order, 1)
    s = np.take_along_axis(station, order, 1)
    r = np.take_along_axis(requested, order, 1)
    live = np.arange(E)[None, :] < n_sessions[:, None]
    sess['arrival'][:, :E] = np.where(live, a, 0)
    sess['departure'][:, :E] = np.where(live, d, 0)
    sess['est_departure'][:, :E] = np.where(live, e, 0)
    sess['station'][:, :E] = np.where(live, s, 0)
    req[:, :E] = np.where(live, r, 0.0)
    moer_day = (np.arange(P) % moer_days).astype(np.int32)
    return n_sessions, sess, req, moer_day
