"""BatchedGMMTraceGenerator: valid episodes (the invariants the reference's per-episode generator
guarantees) and the same distribution as GMMsTraceGenerator (session counts, arrival / duration /
energy moments, station usage)."""
import numpy as np

from sustaingym_amd.event_generation import BatchedGMMTraceGenerator, GMMsTraceGenerator


def test_batched_episodes_are_valid():
    g = BatchedGMMTraceGenerator('caltech', 'Summer 2019', seed=0)
    ns, sess, req, day, mp = g.sample_episodes(2000, stride=64)
    assert ns.max() <= 64 and ns.min() >= 0 and day.min() >= 0 and day.max() < g.num_days_in_date_range
    for e in range(0, 2000, 7):
        s = sess[e, :ns[e]]
        assert np.all(np.diff(s['arrival']) >= 0)                            # arrival-sorted
        assert np.all(s['arrival'] < s['departure']) and np.all(s['arrival'] < s['est_departure'])
        assert np.all((0 <= s['arrival']) & (s['departure'] <= 287) & (s['est_departure'] <= 287))
        assert np.all((0 <= s['station']) & (s['station'] < 54))
        assert np.all((req[e, :ns[e]] >= 0) & (req[e, :ns[e]] <= 100))
        last = {}
        for a, d, st in zip(s['arrival'], s['departure'], s['station']):     # station free on arrival
            assert last.get(st, -1) < a
            last[st] = max(d, last.get(st, -1))
    assert np.all(mp >= 0) and mp.mean() > 1.0


def test_batched_matches_reference_distribution():
    site, period = 'caltech', 'Summer 2019'
    bg = BatchedGMMTraceGenerator(site, period, seed=1)
    ns, sess, req, day, mp = bg.sample_episodes(4000, stride=64)
    ref = GMMsTraceGenerator(site, period)
    rn, ra, rd, rr, rs = [], [], [], [], np.zeros(54)
    np.random.seed(99)
    ref.rng = np.random.default_rng(99)
    ref._gmm_random_state = None          # see test_generator_oracle.py
    for _ in range(400):
        ev = ref._create_events()
        rn.append(len(ev['arrival']))
        ra.extend(ev['arrival']); rd.extend(ev['departure'] - ev['arrival']); rr.extend(ev['requested_energy (kWh)'])
        rs += np.bincount(ev['station'], minlength=54)
    m = np.arange(64)[None, :] < ns[:, None]
    a = sess['arrival'][m]; d = (sess['departure'] - sess['arrival'])[m]; r = req[m]
    bs = np.bincount(sess['station'][m], minlength=54)
    assert abs(ns.mean() - np.mean(rn)) < 0.08 * np.mean(rn)
    assert abs(a.mean() - np.mean(ra)) < 6 and abs(d.mean() - np.mean(rd)) < 6
    assert abs(r.mean() - np.mean(rr)) < 0.08 * np.mean(rr)
    # station usage follows the same empirical distribution (total-variation distance)
    tv = 0.5 * np.abs(bs / bs.sum() - rs / rs.sum()).sum()
    assert tv < 0.08, tv
