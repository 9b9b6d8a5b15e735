#!/usr/bin/env python
"""Generates the golden fixtures of tests/golden/ by RUNNING THE REFERENCE's own data-side code
(container only; imports /root/reference through tools/ref_import.py — see that file for what
is stubbed: containers only, no arithmetic).

    python tests/golden/make_golden.py

Fixtures (data only: inputs = site / period / seed, outputs = event tables, MOER, max_profit):
  real_traces.npz   RealTraceGenerator(site, period, sequential=True): for EVERY day of every
                    default period a CRC32 of the event table returned by the reference's
                    _create_events() and of the MOER matrix get_moer() returns after
                    get_event_queue(); full tables + max_profit for a few days; two full MOER days
  gmm_traces.npz    GMMsTraceGenerator(site, period): for several seeds the first two episodes
                    after set_seed(seed) (tables, simulated day, MOER CRC)
  notebook_golden   the one number recorded in the reference repo
                    (examples/evcharging/env_validation.ipynb cell 24): max_profit = 14.45262
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, ROOT)
from ref_import import reference_generators  # noqa: E402

A_PERS_TO_KWH = (1 / 60) * (208 / 1000) * 5


def table_arrays(df, station_ids, cap=100):
    idx = {s: i for i, s in enumerate(station_ids)}
    arr = df['arrival'].values.astype(np.int16)
    dep = df['departure'].values.astype(np.int16)
    est = df['estimated_departure'].values.astype(np.int16)
    st = np.array([idx[s] for s in df['station_id']], dtype=np.int16)
    req = np.minimum(df['requested_energy (kWh)'].values.astype(np.float64), cap)
    return arr, dep, est, st, req


def crc(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return c


def max_profit(arr, dep, req):
    dur = dep.astype(np.int64) - arr.astype(np.int64)
    return float(np.sum(np.minimum(req, dur * 32 * A_PERS_TO_KWH) * 0.03))   # env.py:422-429


def main():
    eg, ut, lm = reference_generators()
    periods = list(ut.DEFAULT_DATE_RANGES)
    real = {}
    for site in ('caltech', 'jpl'):
        for pi, period in enumerate(periods):
            g = eg.RealTraceGenerator(site, period, sequential=True)
            nd = g.num_days_in_date_range
            ev_crc = np.zeros(nd, np.uint32)
            moer_crc = np.zeros(nd, np.uint32)
            counts = np.zeros(nd, np.int32)
            profits = np.zeros(nd, np.float64)
            for seed in range(nd):
                g.set_seed(seed)
                day = g.day
                df = g._create_events()
                a, d, e, s, r = table_arrays(df, g.station_ids) if len(df) else (np.zeros(0, np.int16),) * 4 + (np.zeros(0),)
                ev_crc[seed] = crc(a, d, e, s, r)
                counts[seed] = len(a)
                profits[seed] = max_profit(a, d, r)
                g.set_seed(seed)
                g.get_event_queue()                      # advances the day (event_generation.py:206)
                m = g.get_moer()                         # MOER of the ADVANCED day (env.py:321-323)
                assert m.shape == (289, 37)
                moer_crc[seed] = crc(m[:, 0].astype(np.float64), m[:, 1:].astype(np.float32))
                if seed in (0, 1, 2, 57, nd - 1):
                    key = f'{site}_{pi}_{seed}'
                    real[key + '_arrival'], real[key + '_departure'] = a, d
                    real[key + '_est'], real[key + '_station'], real[key + '_requested'] = e, s, r
                    real[key + '_day'] = np.array(day.strftime('%Y-%m-%d'))
                    real[key + '_next_day'] = np.array(g.day.strftime('%Y-%m-%d'))
                if site == 'caltech' and pi == 2 and seed in (2, nd - 1):
                    real[f'{site}_{pi}_{seed}_moer'] = m
            real[f'{site}_{pi}_event_crc'] = ev_crc
            real[f'{site}_{pi}_moer_crc'] = moer_crc
            real[f'{site}_{pi}_count'] = counts
            real[f'{site}_{pi}_max_profit'] = profits
            print(site, period, 'days', nd, 'sessions/day max', counts.max())
    # unclaimed variant (use_unclaimed=True) for one site/period
    g = eg.RealTraceGenerator('caltech', periods[0], sequential=True, use_unclaimed=True)
    nd = g.num_days_in_date_range
    unc = np.zeros(nd, np.uint32)
    for seed in range(nd):
        g.set_seed(seed)
        df = g._create_events()
        unc[seed] = crc(*table_arrays(df, g.station_ids)) if len(df) else 0
    real['caltech_0_unclaimed_event_crc'] = unc
    real['notebook_max_profit'] = np.array(14.45262)   # env_validation.ipynb cell 24 (seed 2, Spring 2020)
    np.savez_compressed(os.path.join(HERE, 'real_traces.npz'), **real)

    gmm = {}
    for site in ('caltech', 'jpl'):
        for pi, period in enumerate(periods):
            for seed in (0, 1, 7, 123):
                g = eg.GMMsTraceGenerator(site, period, seed=99)
                g.set_seed(seed)
                for ep in range(2):
                    df = g._create_events()
                    a, d, e, s, r = table_arrays(df, g.station_ids) if len(df) else (np.zeros(0, np.int16),) * 4 + (np.zeros(0),)
                    g._update_day()
                    m = g.get_moer()
                    key = f'{site}_{pi}_{seed}_{ep}'
                    gmm[key + '_arrival'], gmm[key + '_departure'] = a, d
                    gmm[key + '_est'], gmm[key + '_station'], gmm[key + '_requested'] = e, s, r
                    gmm[key + '_day'] = np.array(g.day.strftime('%Y-%m-%d'))
                    gmm[key + '_moer_crc'] = np.uint32(crc(m[:, 0].astype(np.float64), m[:, 1:].astype(np.float32)))
            # constructor path: GMMsTraceGenerator(site, period, seed=s) then first episode
            g = eg.GMMsTraceGenerator(site, period, seed=5)
            gmm[f'{site}_{pi}_ctor5_day'] = np.array(g.day.strftime('%Y-%m-%d'))
            df = g._create_events()
            gmm[f'{site}_{pi}_ctor5_crc'] = np.uint32(crc(*table_arrays(df, g.station_ids)))
        print('gmm', site, 'done')
    np.savez_compressed(os.path.join(HERE, 'gmm_traces.npz'), **gmm)
    for f in ('real_traces.npz', 'gmm_traces.npz'):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
