"""Network descriptors as data: export from an acnportal-shaped object, JSON round trip, precedence of an
exported file in site_str_to_site (utils.py:83-88), and the warning that marks the built-in JPL constraint
set as a provisional stand-in (it is not recoverable from the reference snapshot)."""
import warnings

import numpy as np
import pytest

from sustaingym_amd.network import (ChargingNetwork, ProvisionalNetworkWarning, caltech_acn, jpl_acn,
                                    site_str_to_site)


class _AcnportalShaped:
    """The attributes of acnportal's ChargingNetwork the reference reads (env.py:133-134,373,451,485-493)."""

    def __init__(self, net):
        self.station_ids = list(net.station_ids)
        self.constraint_matrix = net.constraint_matrix.copy()
        self._phase_angles = net.phase_angles.copy()
        self.magnitudes = net.magnitudes.copy()
        self.min_pilot_signals = net.min_pilot_signals
        self.constraint_index = list(net.constraint_names)
        self._voltages = np.full(net.num_stations, 208.0)


@pytest.mark.parametrize('make', [caltech_acn, jpl_acn])
def test_export_round_trip(make, tmp_path, monkeypatch):
    net = make()
    exported = ChargingNetwork.from_acnportal(_AcnportalShaped(net), net.site)
    assert not exported.provisional
    path = tmp_path / f'{net.site}.json'
    exported.to_json(str(path))
    back = ChargingNetwork.from_json(str(path))
    for a, b in ((back.constraint_matrix, net.constraint_matrix), (back.phase_angles, net.phase_angles),
                 (back.magnitudes, net.magnitudes), (back.evse_kind, net.evse_kind)):
        assert np.array_equal(a, b) and a.dtype == b.dtype
    assert back.station_ids == net.station_ids and back.constraint_names == net.constraint_names
    monkeypatch.setenv(f'SUSTAINGYM_AMD_NETWORK_{net.site.upper()}', str(path))
    loaded = site_str_to_site(net.site)
    assert not loaded.provisional and np.array_equal(loaded.constraint_matrix, net.constraint_matrix)


def test_exported_network_must_keep_the_station_order(tmp_path, monkeypatch):
    net = jpl_acn()
    net.station_ids = net.station_ids[::-1]
    path = tmp_path / 'bad.json'
    net.to_json(str(path))
    monkeypatch.setenv('SUSTAINGYM_AMD_NETWORK_JPL', str(path))
    with pytest.raises(ValueError, match='station_ids differ'):
        site_str_to_site('jpl')


def test_evse_kind_mirrors_the_reference_rule():
    """env.py:373-378: `min_pilot_signals[i] == 6` is the AeroVironment EVSE, EVERYTHING else the {0, 8, 16, 24, 32} A one —
    also a FiniteRatesEVSE reported with min_rate = allowable_rates[0] = 0; unheard-of values only warn."""
    from sustaingym_amd.network import EVSE_AV, EVSE_CC
    shaped = _AcnportalShaped(caltech_acn())
    want = np.asarray(caltech_acn().evse_kind)
    mp = np.where(want == EVSE_AV, 6.0, 0.0)                   # ClipperCreek reported as 0 A
    shaped.min_pilot_signals = mp
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        got = ChargingNetwork.from_acnportal(shaped, 'caltech')
    assert np.array_equal(got.evse_kind, want)
    shaped.min_pilot_signals = np.where(want == EVSE_AV, 6.0, 5.0)
    with pytest.warns(UserWarning, match='not 6 A'):
        got = ChargingNetwork.from_acnportal(shaped, 'caltech')
    assert np.array_equal(got.evse_kind, want) and (got.evse_kind == EVSE_CC).any()


def test_provisional_warning(monkeypatch):
    monkeypatch.delenv('SUSTAINGYM_AMD_NETWORK_JPL', raising=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        site_str_to_site('caltech').warn_if_provisional()
        assert not w
        site_str_to_site('jpl').warn_if_provisional()
        assert len(w) == 1 and issubclass(w[0].category, ProvisionalNetworkWarning)
        assert 'from_acnportal' in str(w[0].message)


def test_rollouts_module_has_no_host_policy_loop():
    """The policy runner is the build's own (one batch, evc_rollout); importing it needs no GPU."""
    from sustaingym_amd import rollouts
    assert rollouts.POLICIES == ('greedy', 'random')
    with pytest.raises(ValueError):
        rollouts.PolicyRollout(object.__new__(object), 'mpc')
