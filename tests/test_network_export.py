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


def test_unsupported_evse_rejected():
    shaped = _AcnportalShaped(caltech_acn())
    shaped.min_pilot_signals = np.full(54, 0.0)          # BASIC EVSEs: continuous pilots, not env.py:373's rule
    with pytest.raises(ValueError, match='unsupported EVSE'):
        ChargingNetwork.from_acnportal(shaped, 'caltech')


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
