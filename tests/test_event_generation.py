"""Pins the host-side episode generators (sustaingym_amd/event_generation.py) and the oracle's
max_profit against golden vectors produced by the REFERENCE's own code
(tests/golden/make_golden.py ran RealTraceGenerator / GMMsTraceGenerator / MOERLoader from
/root/reference).  Integer event tables must match bit-exactly."""
import os
import zlib

import numpy as np
import pytest

from oracle import binding as ob
from sustaingym_amd.event_generation import (DEFAULT_DATE_RANGES, GMMsTraceGenerator,
                                             RealTraceGenerator, make_event_table)

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def crc(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return c


def table_crc(ev, cap=100):
    if len(ev['arrival']) == 0:
        return crc(*(np.zeros(0, np.int16),) * 4, np.zeros(0))
    return crc(ev['arrival'].astype(np.int16), ev['departure'].astype(np.int16),
               ev['estimated_departure'].astype(np.int16), ev['station'].astype(np.int16),
               np.minimum(ev['requested_energy (kWh)'].astype(np.float64), cap))


@pytest.fixture(scope='module')
def real():
    return np.load(os.path.join(GOLD, 'real_traces.npz'))


@pytest.fixture(scope='module')
def gmm():
    return np.load(os.path.join(GOLD, 'gmm_traces.npz'))


@pytest.mark.parametrize('site', ['caltech', 'jpl'])
@pytest.mark.parametrize('pi', [0, 1, 2, 3])
def test_real_traces_every_day(real, site, pi):
    """Every day of every packaged period: event table, MOER matrix and max_profit."""
    g = RealTraceGenerator(site, DEFAULT_DATE_RANGES[pi], sequential=True)
    nd = g.num_days_in_date_range
    assert nd == len(real[f'{site}_{pi}_event_crc'])
    for seed in range(nd):
        g.set_seed(seed)
        ev = g._create_events()
        assert table_crc(ev) == real[f'{site}_{pi}_event_crc'][seed], (site, pi, seed)
        assert len(ev['arrival']) == real[f'{site}_{pi}_count'][seed]
        g.set_seed(seed)
        table = g.get_event_table()                    # advances the day
        m = g.get_moer()                               # MOER of the advanced day (env.py:321-323)
        assert crc(m[:, 0], m[:, 1:].astype(np.float32)) == real[f'{site}_{pi}_moer_crc'][seed]
        mp = real[f'{site}_{pi}_max_profit'][seed]
        assert abs(table.max_profit() - mp) <= 1e-12 * max(1.0, abs(mp))
        assert abs(ob.max_profit(table.sessions, table.requested) - mp) <= 1e-12 * max(1.0, abs(mp))


def test_real_traces_full_tables_and_days(real):
    for site in ('caltech', 'jpl'):
        for pi in range(4):
            g = RealTraceGenerator(site, DEFAULT_DATE_RANGES[pi])
            for seed in (0, 1, 2, 57, g.num_days_in_date_range - 1):
                key = f'{site}_{pi}_{seed}'
                g.set_seed(seed)
                assert g.day.strftime('%Y-%m-%d') == str(real[key + '_day'])
                ev = g._create_events()
                assert np.array_equal(ev['arrival'], real[key + '_arrival'])
                assert np.array_equal(ev['departure'], real[key + '_departure'])
                assert np.array_equal(ev['estimated_departure'], real[key + '_est'])
                assert np.array_equal(ev['station'], real[key + '_station'])
                assert np.array_equal(np.minimum(ev['requested_energy (kWh)'], 100), real[key + '_requested'])
                g.get_event_table()
                assert g.day.strftime('%Y-%m-%d') == str(real[key + '_next_day'])


def test_notebook_golden_max_profit(real):
    """The one number recorded in the reference repo (env_validation.ipynb cell 24)."""
    g = RealTraceGenerator('caltech', ('2020-02-01', '2020-05-31'))
    g.set_seed(2)
    table = g.get_event_table()
    assert len(table) == 26
    assert abs(table.max_profit() - float(real['notebook_max_profit'])) < 5e-6
    assert abs(ob.max_profit(table.sessions, table.requested) - 14.45262) < 5e-6
    m = g.get_moer()
    assert np.array_equal(m[:, 0], real['caltech_2_2_moer'][:, 0])
    assert m.dtype == np.float64 and np.array_equal(m, real['caltech_2_2_moer'])       # float64, like the reference's matrix


def test_real_traces_unclaimed(real):
    g = RealTraceGenerator('caltech', DEFAULT_DATE_RANGES[0], use_unclaimed=True)
    for seed in range(g.num_days_in_date_range):
        g.set_seed(seed)
        ev = g._create_events()
        exp = real['caltech_0_unclaimed_event_crc'][seed]
        got = table_crc(ev) if len(ev['arrival']) else 0
        assert got == exp, seed


def test_last_day_wraps(real):
    g = RealTraceGenerator('caltech', 'Spring 2020')
    nd = g.num_days_in_date_range
    g.set_seed(nd - 1)
    g.get_event_table()
    assert g.day == g.date_range[0]
    m = g.get_moer()
    assert np.array_equal(m[:, 0], real[f'caltech_2_{nd - 1}_moer'][:, 0])


@pytest.mark.parametrize('site', ['caltech', 'jpl'])
def test_gmm_traces(gmm, site):
    """GMM episodes: same sklearn/numpy random streams as the reference => identical tables."""
    for pi in range(4):
        for seed in (0, 1, 7, 123):
            g = GMMsTraceGenerator(site, DEFAULT_DATE_RANGES[pi], seed=99)
            g.set_seed(seed)
            for ep in range(2):
                key = f'{site}_{pi}_{seed}_{ep}'
                ev = g._create_events()
                assert np.array_equal(ev['arrival'], gmm[key + '_arrival']), key
                assert np.array_equal(ev['departure'], gmm[key + '_departure']), key
                assert np.array_equal(ev['estimated_departure'], gmm[key + '_est']), key
                assert np.array_equal(ev['station'], gmm[key + '_station']), key
                assert np.array_equal(ev['requested_energy (kWh)'], gmm[key + '_requested']), key
                g._update_day()
                assert g.day.strftime('%Y-%m-%d') == str(gmm[key + '_day'])
                m = g.get_moer()
                assert crc(m[:, 0], m[:, 1:].astype(np.float32)) == int(gmm[key + '_moer_crc'])
        g = GMMsTraceGenerator(site, DEFAULT_DATE_RANGES[pi], seed=5)
        assert g.day.strftime('%Y-%m-%d') == str(gmm[f'{site}_{pi}_ctor5_day'])
        assert table_crc(g._create_events()) == int(gmm[f'{site}_{pi}_ctor5_crc'])


def test_event_table_is_arrival_sorted_and_capped():
    t = make_event_table([5, 1, 5, 0], [9, 8, 7, 6], [9, 9, 9, 9], [0, 1, 2, 3], [10, 200, 30, 40], cap=100)
    assert list(t.sessions['arrival']) == [0, 1, 5, 5]
    assert list(t.sessions['station']) == [3, 1, 0, 2]          # stable
    assert list(t.requested) == [40, 100, 10, 30]


def test_unsupported_options_fail_loudly():
    with pytest.raises(NotImplementedError):
        RealTraceGenerator('caltech', ('2018-11-05', '2018-11-11'))
    with pytest.raises(NotImplementedError):
        GMMsTraceGenerator('caltech', ('2019-05-02', '2019-08-30'))
    with pytest.raises(NotImplementedError):
        RealTraceGenerator('caltech', 'Summer 2019', requested_energy_cap=150)
