#!/usr/bin/env python
"""Builds the packaged derived data of sustaingym_amd from the reference's data files.

Container-only (reads /root/reference through tools/ref_import.py).  Outputs, all numeric
arrays (no reference source, no pickles):

  sustaingym_amd/data/acn_sessions_{site}.npz   every ACN-Data session of the 4 default periods
        (absolute times as UTC epoch seconds, LA-local wall-clock fields, station index,
        requested / delivered kWh, claimed flag)  <- data/evcharging/acn_data/{site}/*.csv.gz
  sustaingym_amd/data/moer_SGIP_CAISO_SCE.npz   5-minute MOER history (float64) + 36 forecasts
        (float64 like the reference's DataFrame; the engine casts to float32 at upload, env.py:390-391) per default period
        <- data/moer/SGIP_CAISO_SCE_*.csv.gz via the reference's load_moer
  sustaingym_amd/data/gmm_{site}.npz            GMM parameters, daily session counts and
        station usage of the 8 pickled models   <- data/evcharging/gmms/{site}/*.pkl
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from ref_import import reference_generators  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'sustaingym_amd', 'data')


def main():
    eg, ut, lm = reference_generators()
    import pandas as pd
    from sustaingym_amd.network import site_str_to_site
    os.makedirs(OUT, exist_ok=True)
    periods = list(ut.DEFAULT_DATE_RANGES)

    for site in ('caltech', 'jpl'):
        net = site_str_to_site(site)
        cols = {k: [] for k in ('period', 'arr_utc', 'dep_utc', 'est_utc', 'arr_min', 'dep_min',
                                'est_min', 'dep_dom', 'est_dom', 'station', 'requested',
                                'delivered', 'claimed')}
        for pi, (a, b) in enumerate(periods):
            df = ut.get_real_events(ut.to_la_dt(a), ut.to_la_dt(b), site)
            for name, col in (('arr', 'arrival'), ('dep', 'departure'), ('est', 'estimated_departure')):
                s = df[col]
                cols[f'{name}_utc'].append((s.dt.tz_convert('UTC').dt.tz_localize(None).astype('datetime64[s]')
                                            .astype('int64')).values)
                cols[f'{name}_min'].append((s.dt.hour * 60 + s.dt.minute).values.astype(np.int16))
                if name != 'arr':
                    cols[f'{name}_dom'].append(s.dt.day.values.astype(np.int8))
            cols['period'].append(np.full(len(df), pi, np.int8))
            cols['station'].append(np.array([net._idx.get(s, -1) for s in df['station_id']], np.int16))
            cols['requested'].append(df['requested_energy (kWh)'].values.astype(np.float64))
            cols['delivered'].append(df['delivered_energy (kWh)'].values.astype(np.float64))
            cols['claimed'].append(df['claimed'].values.astype(np.bool_))
        arrays = {k: np.concatenate(v) for k, v in cols.items()}
        arrays['station_ids'] = np.array(net.station_ids)
        arrays['periods'] = np.array(periods)
        path = os.path.join(OUT, f'acn_sessions_{site}.npz')
        np.savez_compressed(path, **arrays)
        print(path, len(arrays['period']), 'sessions', os.path.getsize(path) // 1024, 'KiB')

        gm = {}
        for pi, (a, b) in enumerate(periods):
            d = ut.load_gmm_model(site, ut.to_la_dt(a), ut.to_la_dt(b), 30)
            g = d[ut.GMM_KEY]
            gm[f'weights_{pi}'] = np.asarray(g.weights_, np.float64)
            gm[f'means_{pi}'] = np.asarray(g.means_, np.float64)
            gm[f'covariances_{pi}'] = np.asarray(g.covariances_, np.float64)
            gm[f'count_{pi}'] = np.asarray(d[ut.COUNT_KEY], np.float64)
            gm[f'station_usage_{pi}'] = np.asarray(d[ut.STATION_USAGE_KEY], np.int64)
            assert g.covariance_type == 'full'
        gm['periods'] = np.array(periods)
        path = os.path.join(OUT, f'gmm_{site}.npz')
        np.savez_compressed(path, **gm)
        print(path, os.path.getsize(path) // 1024, 'KiB')

    moer = {}
    for pi, (a, b) in enumerate(periods):
        loader = lm.MOERLoader(ut.to_la_dt(a), ut.to_la_dt(b), 'SGIP_CAISO_SCE', 'sustaingym/data/moer')
        df = loader.df
        t = df.index.tz_convert('UTC').tz_localize(None).astype('datetime64[s]').astype('int64').values
        assert np.all(np.diff(t) > 0)
        gaps = np.sum(np.diff(t) != 300)
        assert gaps == 0, 'MOER series must be gap-free 5-minute data'
        moer[f't0_{pi}'] = np.int64(t[0])
        moer[f'hist_{pi}'] = np.ascontiguousarray(df.values[:, 0], dtype=np.float64)
        moer[f'fcst_{pi}'] = np.ascontiguousarray(df.values[:, 1:], dtype=np.float64)   # MOERLoader.retrieve returns float64 (load_moer.py:364-377)
        print('moer period', pi, df.shape, 'gaps', gaps)
    moer['periods'] = np.array(periods)
    path = os.path.join(OUT, 'moer_SGIP_CAISO_SCE.npz')
    np.savez_compressed(path, **moer)
    print(path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
