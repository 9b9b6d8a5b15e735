"""Episode generation (reset-time, host side): event tables and MOER matrices.

Restates, on numpy arrays, the reference's trace generators
(sustaingym/envs/evcharging/event_generation.py) and MOER loader
(sustaingym/data/load_moer.py:346-377) with the same constructor arguments, attributes and
random-number consumption, so that ``reset(seed=s)`` yields the same episode as the reference.
Instead of an ``acnportal`` ``EventQueue`` of Python objects the generators return an
:class:`EventTable` — the arrival-sorted session arrays the HIP engine consumes
(include/evcharge.h ``evc_session``).  Every function cites the reference lines it follows;
tests/test_event_generation.py checks them against golden tables produced by the reference's
own code (tests/golden/make_golden.py).

Data comes from ``sustaingym_amd/data/*.npz`` (built by tools/build_data.py from the reference's
packaged ACN-Data / SGIP MOER / GMM files).  Custom date ranges must lie inside one of the four
packaged periods (the reference would otherwise call the ACN-Data web API / retrain a GMM).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from datetime import datetime, timedelta, timezone
from functools import lru_cache

import numpy as np

from ._lib import SESSION_DTYPE
from .network import site_str_to_site

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')

DATE_FORMAT = '%Y-%m-%d'                   # utils.py:33
MINS_IN_DAY = 1440                          # utils.py:37
REQ_ENERGY_SCALE = 100                      # utils.py:41
# utils.py:29,78-80: ``datetime.replace(tzinfo=pytz.timezone('America/Los_Angeles'))`` attaches
# pytz's *LMT* offset (-7:53), not PST/PDT.  This quirk defines where a "day" starts; keep it.
AM_LA_LMT = timezone(timedelta(hours=-7, minutes=-53))
_LMT_OFFSET_S = 7 * 3600 + 53 * 60

DEFAULT_DATE_RANGES = (                     # utils.py:48-53
    ('2019-05-01', '2019-08-31'),
    ('2019-09-01', '2019-12-31'),
    ('2020-02-01', '2020-05-31'),
    ('2021-05-01', '2021-08-31'),
)
DEFAULT_PERIOD_TO_RANGE = {                 # utils.py:55-64
    'Summer 2019': DEFAULT_DATE_RANGES[0], 'Pre-COVID-19 Summer': DEFAULT_DATE_RANGES[0],
    'Fall 2019': DEFAULT_DATE_RANGES[1], 'Pre-COVID-19 Fall': DEFAULT_DATE_RANGES[1],
    'Spring 2020': DEFAULT_DATE_RANGES[2], 'In-COVID-19': DEFAULT_DATE_RANGES[2],
    'Summer 2021': DEFAULT_DATE_RANGES[3], 'Post-COVID-19': DEFAULT_DATE_RANGES[3],
}


def to_la_dt(s: str) -> datetime:
    """utils.py:78-80."""
    return datetime.strptime(s, DATE_FORMAT).replace(tzinfo=AM_LA_LMT)


def _epoch(dt: datetime) -> int:
    return int(dt.timestamp())


def _period_index(date_range: tuple[datetime, datetime]) -> int:
    """Packaged period containing the date range (utils.py:193-196)."""
    for i, (a, b) in enumerate(DEFAULT_DATE_RANGES):
        if to_la_dt(a) <= date_range[0] and date_range[1] <= to_la_dt(b) + timedelta(days=1):
            return i
    raise NotImplementedError(
        f'date range {date_range[0]:%Y-%m-%d}..{date_range[1]:%Y-%m-%d} is not inside a packaged '
        'period; the reference would fetch it from the ACN-Data API (no network here)')


@lru_cache(maxsize=None)
def _load_npz(name: str):
    return dict(np.load(os.path.join(_DATA, name), allow_pickle=False))


@dataclass
class EventTable:
    """One episode: sessions in arrival order (stable), as the engine consumes them."""
    sessions: np.ndarray      # SESSION_DTYPE [E]
    requested: np.ndarray     # float64 [E], already capped (event_generation.py:169-170)

    def __len__(self) -> int:
        return len(self.sessions)

    @property
    def arrival(self):
        return self.sessions['arrival']

    @property
    def departure(self):
        return self.sessions['departure']

    def max_profit(self) -> float:
        """env.py:422-429."""
        a_pers_to_kwh = (1 / 60) * (208 / 1000) * 5
        dur = (self.sessions['departure'].astype(np.int64) - self.sessions['arrival'].astype(np.int64))
        max_kwh = dur * 32 * a_pers_to_kwh
        return float(np.sum(np.minimum(self.requested, max_kwh) * (0.15 * 0.20)))

    def avg_plugin_time(self) -> float:
        """env.py:418-420."""
        dur = self.sessions['departure'].astype(np.int64) - self.sessions['arrival'].astype(np.int64)
        return float(np.mean(dur)) if len(dur) else float('nan')


def make_event_table(arrival, departure, est_departure, station, requested, cap: float) -> EventTable:
    arrival = np.asarray(arrival, dtype=np.int64)
    order = np.argsort(arrival, kind='stable')
    s = np.empty(len(arrival), dtype=SESSION_DTYPE)
    s['arrival'] = arrival[order]
    s['departure'] = np.asarray(departure, dtype=np.int64)[order]
    s['est_departure'] = np.asarray(est_departure, dtype=np.int64)[order]
    s['station'] = np.asarray(station, dtype=np.int64)[order]
    req = np.minimum(np.asarray(requested, dtype=np.float64)[order], cap)  # :169-170
    return EventTable(s, req)


class MOERLoader:
    """sustaingym/data/load_moer.py:346-377 on the packaged arrays."""

    def __init__(self, starttime: datetime, endtime: datetime, ba: str = 'SGIP_CAISO_SCE',
                 save_dir: str | None = None):
        if ba != 'SGIP_CAISO_SCE':
            raise NotImplementedError('only the SGIP_CAISO_SCE series (Caltech/JPL) is packaged')
        self._pi = _period_index((starttime, endtime))
        d = _load_npz('moer_SGIP_CAISO_SCE.npz')
        self._t0 = int(d[f't0_{self._pi}'])
        self._hist = d[f'hist_{self._pi}']
        self._fcst = d[f'fcst_{self._pi}']

    def retrieve(self, dt: datetime) -> np.ndarray:
        """load_moer.py:364-377: rows with dt <= index < dt + 1 day + 5 min -> [289, 37].

        float64 throughout, like the reference's DataFrame values (``info['moer']``); the engine casts
        the observation's columns to float32 at upload (env.py:140,390-391)."""
        start = -(-(_epoch(dt) - self._t0) // 300)        # first 5-minute mark >= dt
        if start < 0 or start + 289 > len(self._hist):
            raise ValueError(f'MOER data does not cover {dt}')
        out = np.empty((289, 37), dtype=np.float64)
        out[:, 0] = self._hist[start:start + 289]
        out[:, 1:] = self._fcst[start:start + 289]
        return out


class AbstractTraceGenerator:
    """event_generation.py:27-218 (same arguments / attributes)."""
    TIME_STEP_DURATION = 5          # :55
    MAX_STEPS_OF_TRACE = 288        # :57
    BATTERY_CAPACITY = 100          # :60
    MAX_POWER = 100                 # :62
    BA_CALTECH_JPL = 'SGIP_CAISO_SCE'

    def __init__(self, site: str, date_period, requested_energy_cap: float = 100,
                 seed: int | None = None):
        self.site = site
        self.network = site_str_to_site(site)
        self.station_ids = self.network.station_ids
        self.num_stations = len(self.station_ids)
        if isinstance(date_period, str):
            self.date_range_str = DEFAULT_PERIOD_TO_RANGE[date_period]      # :79-83
        else:
            self.date_range_str = tuple(date_period)
        self.date_range = tuple(to_la_dt(s) for s in self.date_range_str)   # :86
        self.num_days_in_date_range = (self.date_range[1] - self.date_range[0]).days + 1  # :89
        if requested_energy_cap > self.BATTERY_CAPACITY:
            raise NotImplementedError(
                'requested_energy_cap above the 100 kWh battery capacity is not supported by the '
                'HIP engine (battery headroom would differ from the remaining demand)')
        self.requested_energy_cap = requested_energy_cap
        self.moer_loader = MOERLoader(self.date_range[0], self.date_range[1], self.BA_CALTECH_JPL)
        self.rng = np.random.default_rng(seed=seed)                         # :99
        self.day: datetime

    def site__repr__(self) -> str:
        return ('JPL' if self.site == 'jpl' else self.site.capitalize()) + ' garage'

    def date_range__repr__(self) -> str:
        return f'({self.date_range_str[0]} to {self.date_range_str[1]})'

    def _update_day(self) -> None:                                          # :117-119
        self.day = self.date_range[0] + timedelta(days=int(self.rng.choice(self.num_days_in_date_range)))

    def set_seed(self, seed: int | None) -> None:                           # :121-123
        self.rng = np.random.default_rng(seed=seed)

    def _create_events(self) -> dict[str, np.ndarray]:
        raise NotImplementedError

    def get_event_table(self) -> EventTable:
        """Counterpart of get_event_queue (:149-207): builds the episode, then updates the day."""
        ev = self._create_events()
        table = make_event_table(ev['arrival'], ev['departure'], ev['estimated_departure'],
                                 ev['station'], ev['requested_energy (kWh)'], self.requested_energy_cap)
        self._update_day()                                                  # :206
        return table

    def get_event_queue(self):
        """Reference signature (:149): ``(events, evs, num_plugin)``.  ``events`` and ``evs`` are
        the same :class:`EventTable` (there are no per-EV Python objects in this engine)."""
        table = self.get_event_table()
        return table, table, len(table)

    def get_moer(self) -> np.ndarray:                                       # :209-218
        return self.moer_loader.retrieve(self.day)


class RealTraceGenerator(AbstractTraceGenerator):
    """event_generation.py:221-328."""

    def __init__(self, site: str, date_period, sequential: bool = True, use_unclaimed: bool = False,
                 requested_energy_cap: float = 100, seed: int | None = None):
        super().__init__(site, date_period, requested_energy_cap, seed)
        self.sequential = sequential
        if sequential:                                                       # :255-260
            if seed is None:
                seed = 0
            self.set_seed(seed)
        else:
            self._update_day()
        self.use_unclaimed = use_unclaimed
        d = _load_npz(f'acn_sessions_{site}.npz')
        pi = _period_index(self.date_range)
        lo, hi = _epoch(self.date_range[0]), _epoch(self.date_range[1] + timedelta(days=1))
        keep = (d['period'] == pi) & (lo <= d['arr_utc']) & (d['arr_utc'] <= hi)   # utils.py:202
        self._ev = {k: d[k][keep] for k in ('arr_utc', 'dep_utc', 'est_utc', 'arr_min', 'dep_min',
                                            'est_min', 'dep_dom', 'est_dom', 'station', 'requested',
                                            'delivered', 'claimed')}

    def __repr__(self) -> str:
        return (f'Real trace generator for {self.site__repr__()} {self.date_range__repr__()}\n'
                f'Sequential = {self.sequential}, Use unclaimed = {self.use_unclaimed}\n'
                f'Current day: {self.day.strftime(DATE_FORMAT)}')

    def set_seed(self, seed: int | None) -> None:                            # :273-282
        if self.sequential:
            if seed is None:
                seed = 0
            self.day = self.date_range[0] + timedelta(days=seed % self.num_days_in_date_range)
        else:
            super().set_seed(seed)

    def _update_day(self) -> None:                                           # :284-291
        if self.sequential:
            self.day += timedelta(days=1)
            if self.day > self.date_range[1]:
                self.day = self.date_range[0]
        else:
            super()._update_day()

    def _create_events(self) -> dict[str, np.ndarray]:                       # :293-328
        e = self._ev
        day0 = _epoch(self.day)
        m = (day0 <= e['arr_utc']) & (e['arr_utc'] < day0 + 86400)           # :301-302
        if not self.use_unclaimed:
            m &= e['claimed']                                                # :303-304
        m &= e['station'] >= 0                                               # :307
        # :313-316 the later of departure / estimated departure must fall on day.day (LA date)
        later_dom = np.where(e['dep_utc'] >= e['est_utc'], e['dep_dom'], e['est_dom'])
        m &= later_dom == self.day.day
        arr = e['arr_min'][m].astype(np.int64) // self.TIME_STEP_DURATION    # :323-324
        dep = e['dep_min'][m].astype(np.int64) // self.TIME_STEP_DURATION
        est = e['est_min'][m].astype(np.int64) // self.TIME_STEP_DURATION
        ok = est > arr                                                       # :327
        return {'arrival': arr[ok], 'departure': dep[ok], 'estimated_departure': est[ok],
                'station': e['station'][m][ok].astype(np.int64),
                'requested_energy (kWh)': e['requested'][m][ok],
                'delivered_energy (kWh)': e['delivered'][m][ok]}


class GMMsTraceGenerator(AbstractTraceGenerator):
    """event_generation.py:331-515."""
    ARRCOL, DEPCOL, ESTCOL, EREQCOL = 0, 1, 2, 3

    def __init__(self, site: str, date_period, n_components: int = 30,
                 requested_energy_cap: float = 100, seed: int | None = None):
        super().__init__(site, date_period, requested_energy_cap, seed)
        if n_components != 30:
            raise NotImplementedError('only the packaged 30-component GMMs are available')
        self.n_components = n_components
        try:
            pi = DEFAULT_DATE_RANGES.index(tuple(self.date_range_str))
        except ValueError as exc:
            raise NotImplementedError('GMMs exist only for the four default periods; the reference '
                                      'would train a new one') from exc
        d = _load_npz(f'gmm_{site}.npz')
        self.weights_ = d[f'weights_{pi}']
        self.means_ = d[f'means_{pi}']
        self.covariances_ = d[f'covariances_{pi}']
        self.cnt = d[f'count_{pi}']
        self.station_usage = d[f'station_usage_{pi}']
        self.set_seed(seed)                                                  # :403-404
        self._update_day()

    def __repr__(self) -> str:
        return (f'{self.n_components}-component GMM-based trace generator for '
                f'{self.site__repr__()} {self.date_range__repr__()}')

    def set_seed(self, seed: int | None) -> None:                            # :411-414
        super().set_seed(seed)
        self._gmm_random_state = seed     # gmm.set_params(random_state=seed)

    def _gmm_sample(self, n_samples: int) -> np.ndarray:
        """sklearn ``GaussianMixture.sample`` (covariance_type='full'): a fresh
        ``check_random_state(random_state)`` per call, multinomial component counts, then
        ``multivariate_normal`` per component, stacked in component order."""
        if self._gmm_random_state is None:
            rs = np.random.mtrand._rand
        else:
            rs = np.random.RandomState(self._gmm_random_state)
        comp = rs.multinomial(n_samples, self.weights_)
        return np.vstack([rs.multivariate_normal(mean, cov, int(k))
                          for mean, cov, k in zip(self.means_, self.covariances_, comp)])

    def _sample(self, n: int, oversample_factor: float = 0.2) -> np.ndarray:  # :416-463
        if n == 0:
            return np.empty((0, 4))
        all_samples, num = [], 0
        while num < n:
            s = self._gmm_sample(int(n * (1 + oversample_factor)))
            s = s[(0 <= s[:, 0]) & (s[:, 1] < 1) & (s[:, 2] < 1) & (s[:, 3] >= 0)]
            s[:, [0, 1, 2]] = MINS_IN_DAY * s[:, [0, 1, 2]] // self.TIME_STEP_DURATION
            s = s[(s[:, 0] < s[:, 1]) & (s[:, 0] < s[:, 2])]
            s[:, 3] *= REQ_ENERGY_SCALE
            all_samples.append(s)
            num += len(s)
        return np.concatenate(all_samples, axis=0)[:n]

    def _create_events(self) -> dict[str, np.ndarray]:                       # :465-515
        n = int(self.rng.choice(self.cnt))
        samples = self._sample(n)
        arrival = samples[:, 0].astype(int)
        # :490 ``events.sort_values('arrival')``: pandas' default (numpy 'quicksort') argsort
        order = np.argsort(arrival, kind='quicksort')
        arrival = arrival[order]
        departure = samples[:, 1].astype(int)[order]
        est = samples[:, 2].astype(int)[order]
        req = np.clip(samples[:, 3], 0, self.requested_energy_cap)[order]
        station_cnts = self.station_usage / self.station_usage.sum()
        station_dep = np.full(self.num_stations, -1, dtype=np.int32)
        station = np.full(n, -1, dtype=np.int64)
        for i in range(n):
            avail = np.where(station_dep < arrival[i])[0]
            if len(avail) == 0:
                continue
            csum = station_cnts[avail].sum()
            if csum <= 1e-5:
                idx = self.rng.choice(avail)
            else:
                idx = self.rng.choice(avail, p=station_cnts[avail] / csum)
            station_dep[idx] = max(departure[i], station_dep[idx])
            station[i] = idx
        keep = station >= 0
        return {'arrival': arrival[keep], 'departure': departure[keep],
                'estimated_departure': est[keep], 'station': station[keep],
                'requested_energy (kWh)': req[keep]}


class BatchedGMMTraceGenerator:
    """Vectorised episode sampler for large batches of environments (host side, numpy).

    Same model as :class:`GMMsTraceGenerator` — the packaged 30-component GMM over (arrival,
    departure, estimated departure, requested energy), the empirical daily session counts and the
    availability-weighted station assignment of event_generation.py:416-515 — but all
    ``count`` episodes are drawn at once from ONE random stream, so the episodes are
    *distributionally* equivalent to the reference's, not stream-identical (use
    ``GMMsTraceGenerator`` per environment when seed-for-seed reproduction of the reference matters;
    it costs ~2 ms per episode, this sampler ~0.1 ms).  The weighted station choice is an
    exponential race (argmin of Exp(1)/p_i over the free EVSEs), which samples exactly
    ``rng.choice(avail, p=p/p.sum())``.
    """
    TIME_STEP_DURATION = 5

    def __init__(self, site: str, date_period, requested_energy_cap: float = 100, seed: int | None = None):
        self._g = GMMsTraceGenerator(site, date_period, requested_energy_cap=requested_energy_cap, seed=seed)
        self.site = site
        self.date_range_str = self._g.date_range_str
        self.date_range = self._g.date_range
        self.num_days_in_date_range = self._g.num_days_in_date_range
        self.requested_energy_cap = requested_energy_cap
        self.moer_loader = self._g.moer_loader
        self.num_stations = self._g.num_stations
        self.rng = np.random.default_rng(seed)
        self._chol = np.linalg.cholesky(self._g.covariances_)
        p = self._g.station_usage / self._g.station_usage.sum()
        self._never = p <= 0                                 # EVSEs never used in the period
        self._invp = (1.0 / np.where(self._never, 1.0, p)).astype(np.float32)
        self._p32 = p.astype(np.float32)

    def set_seed(self, seed: int | None) -> None:
        self.rng = np.random.default_rng(seed)

    def sample_episodes(self, count: int, stride: int = 128, chunk: int = 4096):
        """Returns ``(n_sessions[count], sessions[count, stride], requested[count, stride],
        day_index[count], max_profit[count])``; ``day_index`` counts days from ``date_range[0]``."""
        if count > chunk:       # cache-sized blocks: the station loop touches [count, n] arrays 84 times
            sizes = [min(chunk, count - i) for i in range(0, count, chunk)]
            rngs = self.rng.spawn(len(sizes))       # one child stream per block: thread-count independent
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(len(sizes), os.cpu_count() or 1, 32)) as pool:
                parts = list(pool.map(lambda a: self._sample_block(a[0], stride, a[1]), zip(sizes, rngs)))
            return tuple(np.concatenate([p[j] for p in parts]) for j in range(5))
        return self._sample_block(count, stride, self.rng)

    def _draw(self, shape: tuple[int, int], rng):
        comp = rng.choice(len(self._g.weights_), size=shape, p=self._g.weights_)
        z = rng.standard_normal(shape + (4,))
        x = np.empty_like(z)
        for c in range(len(self._g.weights_)):
            sel = comp == c
            x[sel] = self._g.means_[c] + z[sel] @ self._chol[c].T
        return comp, x

    def _sample_block(self, count: int, stride: int, rng):
        g, n = self._g, self.num_stations
        want = rng.choice(g.cnt, size=count).astype(np.int64)                       # :479
        want = np.minimum(want, stride)
        E = int(max(1, want.max())) if count else 1
        # :416-463 incl. the reference's surplus cut: every round draws int(1.2 n) samples that
        # sklearn returns stacked in component order, and only the first n accepted are kept.
        m = (want * (1 + 0.2)).astype(np.int64)
        acc = np.zeros((4, count, E))
        have = np.zeros(count, dtype=np.int64)
        K = len(g.weights_)
        for _ in range(16):
            idx = np.nonzero(have < want)[0]
            if len(idx) == 0:
                break
            mi = m[idx]
            B = int(mi.max())
            comp, x = self._draw((len(idx), B), rng)
            inr = np.arange(B)[None, :] < mi[:, None]
            order = np.argsort(np.where(inr, comp, K), axis=1, kind='stable')
            x = np.take_along_axis(x, order[..., None], 1)
            ok = np.take_along_axis(inr, order, 1)
            ok &= (0 <= x[..., 0]) & (x[..., 1] < 1) & (x[..., 2] < 1) & (x[..., 3] >= 0)    # :441-444
            t = np.floor(MINS_IN_DAY * x[..., :3] / self.TIME_STEP_DURATION)             # :447-449
            ok &= (t[..., 0] < t[..., 1]) & (t[..., 0] < t[..., 2])                       # :452-455
            pos = have[idx, None] + np.cumsum(ok, axis=1) - 1
            ok &= pos < want[idx, None]                                                   # first n accepted
            r, c = np.nonzero(ok)
            for k in range(3):
                acc[k, idx[r], pos[r, c]] = t[r, c, k]
            acc[3, idx[r], pos[r, c]] = np.clip(x[r, c, 3] * REQ_ENERGY_SCALE, 0, self.requested_energy_cap)  # :458,486
            have[idx] += ok.sum(axis=1)
        want = have
        filled = np.arange(E)[None, :] < want[:, None]
        order = np.argsort(np.where(filled, acc[0], 1e9), axis=1, kind='stable')       # :490
        take = lambda a: np.take_along_axis(a, order, 1)
        arr = take(acc[0]).astype(np.int64)
        dep = take(acc[1]).astype(np.int64)
        est = take(acc[2]).astype(np.int64)
        req = take(acc[3])
        E = arr.shape[1]
        live = np.arange(E)[None, :] < want[:, None]
        station = np.full((count, E), -1, dtype=np.int64)
        station_dep = np.full((count, n), -1, dtype=np.int64)
        rows = np.arange(count)
        for j in range(int(want.max()) if count else 0):                              # :499-511
            avail = station_dep < arr[:, j:j + 1]
            psum = np.where(avail, self._p32[None, :], np.float32(0)).sum(axis=1)
            race = rng.standard_exponential((count, n), dtype=np.float32)
            weighted = psum[:, None] > 1e-5                               # :504-508 (else uniform)
            race *= np.where(weighted, self._invp[None, :], np.float32(1))
            race[~avail | (weighted & self._never[None, :])] = np.inf
            pick = np.argmin(race, axis=1)
            okj = live[:, j] & np.isfinite(race[rows, pick])
            station[:, j] = np.where(okj, pick, -1)
            station_dep[rows, pick] = np.where(okj, np.maximum(dep[:, j], station_dep[rows, pick]),
                                               station_dep[rows, pick])
        keep = station >= 0                                                            # :514
        order2 = np.argsort(~keep, axis=1, kind='stable')
        n_sessions = keep.sum(axis=1).astype(np.int32)
        take2 = lambda a: np.take_along_axis(a, order2, 1)
        sess = np.zeros((count, stride), dtype=SESSION_DTYPE)
        reqo = np.zeros((count, stride), dtype=np.float64)
        mask = np.arange(E)[None, :] < n_sessions[:, None]
        sess['arrival'][:, :E] = np.where(mask, take2(arr), 0)
        sess['departure'][:, :E] = np.where(mask, take2(dep), 0)
        sess['est_departure'][:, :E] = np.where(mask, take2(est), 0)
        sess['station'][:, :E] = np.where(mask, take2(station), 0)
        reqo[:, :E] = np.where(mask, take2(req), 0.0)
        day = rng.integers(self.num_days_in_date_range, size=count).astype(np.int32)   # :117-119
        a_pers = (1 / 60) * (208 / 1000) * 5
        dur = (sess['departure'].astype(np.int64) - sess['arrival'].astype(np.int64))[:, :E]
        max_profit = np.where(mask, np.minimum(reqo[:, :E], dur * 32 * a_pers) * 0.03, 0.0).sum(axis=1)
        return n_sessions, sess, reqo, day, max_profit


def gmm_device_tables(site: str, date_period, requested_energy_cap: float = 100) -> dict:
    """Tables of the on-device episode generator (include/evcharge.h ``evc_gmm_desc``): the
    packaged GMM of the period with cumulative weights and Cholesky factors precomputed, the
    empirical daily session counts (event_generation.py:479) and the integer EVSE usage counts
    (:497)."""
    g = GMMsTraceGenerator(site, date_period, requested_energy_cap=requested_energy_cap)
    cum = np.cumsum(g.weights_)
    cum[-1] = 1.0
    usage = np.asarray(g.station_usage)
    assert np.all(usage == np.round(usage)) and usage.min() >= 0 and usage.sum() < 2 ** 31
    return {
        'cum_weights': cum, 'means': np.ascontiguousarray(g.means_, dtype=np.float64),
        'chol': np.ascontiguousarray(np.linalg.cholesky(g.covariances_)),
        'daily_counts': np.asarray(g.cnt, dtype=np.int32),
        'station_usage': usage.astype(np.uint32), 'num_days': g.num_days_in_date_range,
        'requested_energy_cap': float(requested_energy_cap),
    }


class DeviceGMMTraceGenerator:
    """Handle for on-device episode generation (``evc_generate_episodes``): names the site, the
    date period and the seed of the counter-based random stream; holds the model tables and the
    MOER loader.  No sampling happens on the host — pass it to ``EVChargingVectorEnv`` (with
    ``num_envs``), which fills its episode bank on the GPU at every reset / autoreset boundary.
    Episodes follow ``GMMsTraceGenerator``'s distribution (same GMM, rejection rules, surplus cut,
    availability-weighted EVSE choice); episode ``e`` of seed ``s`` is reproducible bit for bit."""

    def __init__(self, site: str, date_period, requested_energy_cap: float = 100, seed: int | None = None):
        g = GMMsTraceGenerator(site, date_period, requested_energy_cap=requested_energy_cap)
        self.site = site
        self.date_range_str = g.date_range_str
        self.date_range = g.date_range
        self.num_days_in_date_range = g.num_days_in_date_range
        self.requested_energy_cap = requested_energy_cap
        self.moer_loader = g.moer_loader
        self.num_stations = g.num_stations
        self.tables = gmm_device_tables(site, date_period, requested_energy_cap)
        self.set_seed(seed)

    def set_seed(self, seed: int | None) -> None:
        if seed is None:
            seed = int(np.random.SeedSequence().generate_state(2, dtype=np.uint32).view(np.uint64)[0])
        self.seed = int(seed) & (2 ** 64 - 1)
        self.next_episode = 0


class RealTraceBank:
    """Every day of a packaged period as one resident episode bank (``RealTraceGenerator`` with
    ``sequential=True``, the reference default, for a whole batch of environments).

    Pass it to ``EVChargingVectorEnv`` (with ``num_envs``): the tables of all ``D`` days are uploaded once,
    environment ``i`` starts on day ``(seed + i) mod D`` (``RealTraceGenerator.set_seed``,
    event_generation.py:273-282) and moves to the next day at every episode boundary inside the kernel
    (``_update_day`` :284-291, cycling at the end of the period) — no host work at boundaries.  As in
    the reference the MOER of an episode is that of the *advanced* day (env.py:321-323 calls
    ``get_moer`` after ``get_event_queue``).

    Deviation, on purpose: the reference's ``reset()`` re-seeds its generator with ``None`` at every
    un-seeded reset (env.py:314), which rewinds a sequential generator to the first day
    (event_generation.py:278-281) — under a VectorEnv autoreset every episode after the first would replay
    day 0.  The bank implements the documented intent (``_update_day``: "increments day"); an episode on
    day ``d`` equals the reference's ``reset(seed=d)``.  ``EVChargingVectorEnv`` fed with one
    ``RealTraceGenerator`` per environment keeps the reference's literal behaviour."""

    def __init__(self, site: str, date_period, use_unclaimed: bool = False, requested_energy_cap: float = 100,
                 max_sessions: int = 128):
        g = RealTraceGenerator(site, date_period, sequential=True, use_unclaimed=use_unclaimed,
                               requested_energy_cap=requested_energy_cap)
        self.site = site
        self.date_range_str = g.date_range_str
        self.date_range = g.date_range
        self.num_days_in_date_range = D = g.num_days_in_date_range
        self.requested_energy_cap = requested_energy_cap
        self.moer_loader = g.moer_loader
        self.num_stations = g.num_stations
        self.n_sessions = np.zeros(D, np.int32)
        self.sessions = np.zeros((D, max_sessions), dtype=SESSION_DTYPE)
        self.requested = np.zeros((D, max_sessions))
        self.max_profit = np.zeros(D)
        for d in range(D):
            g.set_seed(d)                                   # day = first day + d
            t = g.get_event_table()
            assert len(t) <= max_sessions, f'day {d} has {len(t)} sessions'
            self.n_sessions[d] = len(t)
            self.sessions[d, :len(t)] = t.sessions
            self.requested[d, :len(t)] = t.requested
            self.max_profit[d] = t.max_profit()
        self.moer_day = ((np.arange(D) + 1) % D).astype(np.int32)     # MOER of the advanced day
