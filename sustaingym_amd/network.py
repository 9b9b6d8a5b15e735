"""Charging-network descriptors (data, not code).

The reference obtains its network from the un-vendored dependency acnportal
(``sustaingym/envs/evcharging/utils.py:83-88`` -> ``acns.network.sites.caltech_acn()`` /
``jpl_acn()``) and reads from it only

* ``cn.station_ids``            (order of the action / observation vectors, env.py:133-134)
* ``cn.constraint_matrix``      (real ``[m, n]``, env.py:486)
* ``cn._phase_angles``          (degrees ``[n]``, env.py:485)
* ``cn.magnitudes``             (amps ``[m]``, env.py:451,493)
* ``cn.min_pilot_signals``      (6 => AeroVironment, 8 => ClipperCreek, env.py:373)

This module restates exactly those fields.

Provenance
----------
* Caltech **station order**: recovered from in-tree artefacts of the reference (the
  ``station_usage`` vectors in the GMM pickles are indexed by ``cn.station_ids``,
  train_gmm_model.py:165-178) and identical to acnportal's registration order
  ``AB_ids (10 + AV pod 8 + CC pod 8), BC_ids (14), CA_ids (14)``.
* Caltech **constraints**: acnportal 0.3.x ``caltech_acn`` (published; restated from memory,
  *parity unpinned*): delta-connected EVSEs at 208 V with phase angles AB=+30, BC=-90,
  CA=+150 degrees; line currents ``I3a = AB - CA``, ``I3b = BC - AB``, ``I3c = CA - BC``
  (secondary side, limit ``150 kVA / 3 / 120 V``); primary side ``I2a = (I3a - I3c)/4`` etc.
  (limit ``150 kVA / 3 / 277 V``); two 80 A pod breakers.  Constraint order as added by
  acnportal: Secondary A,B,C, Primary A,B,C, CC Pod, AV Pod.
* JPL: station order recovered the same way (fully determined).  The JPL constraint set is
  **not** recoverable from /root/reference and is shipped as a clearly-marked PROVISIONAL
  topology (``provisional=True``); see DESIGN.md.
"""
from __future__ import annotations

import json
import os
import warnings
from dataclasses import dataclass, field

import numpy as np

class ProvisionalNetworkWarning(UserWarning):
    """A network whose constraint set is NOT acnportal's (today: the built-in JPL topology) is in use."""


EVSE_AV = 0  # AeroVironment: allowable pilots {0} U {6..32} A, min pilot 6
EVSE_CC = 1  # ClipperCreek: allowable pilots {0, 8, 16, 24, 32} A, min pilot 8

MAX_PILOT = 32.0  # env.py:100 ACTION_SCALE_FACTOR


@dataclass
class ChargingNetwork:
    """The subset of ``acnportal.acnsim.ChargingNetwork`` the reference env reads."""
    site: str
    station_ids: list[str]
    constraint_matrix: np.ndarray   # [m, n] float64
    phase_angles: np.ndarray        # [n] degrees (cn._phase_angles)
    magnitudes: np.ndarray          # [m] amps
    constraint_names: list[str]
    evse_kind: np.ndarray           # [n] uint8 (EVSE_AV / EVSE_CC)
    voltage: float = 208.0
    provisional: bool = False
    _idx: dict = field(default_factory=dict, repr=False)

    def __post_init__(self) -> None:
        self.constraint_matrix = np.ascontiguousarray(self.constraint_matrix, dtype=np.float64)
        self.phase_angles = np.ascontiguousarray(self.phase_angles, dtype=np.float64)
        self.magnitudes = np.ascontiguousarray(self.magnitudes, dtype=np.float64)
        self.evse_kind = np.ascontiguousarray(self.evse_kind, dtype=np.uint8)
        m, n = self.constraint_matrix.shape
        assert n == len(self.station_ids) == len(self.phase_angles) == len(self.evse_kind)
        assert m == len(self.magnitudes) == len(self.constraint_names)
        self._idx = {s: i for i, s in enumerate(self.station_ids)}

    # --- acnportal-compatible read accessors used by the reference ---
    @property
    def _phase_angles(self) -> np.ndarray:  # env.py:485
        return self.phase_angles

    @property
    def min_pilot_signals(self) -> np.ndarray:  # env.py:373
        return np.where(self.evse_kind == EVSE_AV, 6.0, 8.0)

    @property
    def max_pilot_signals(self) -> np.ndarray:
        return np.full(len(self.station_ids), MAX_PILOT)

    @property
    def num_stations(self) -> int:
        return len(self.station_ids)

    def station_index(self, station_id: str) -> int:
        return self._idx[station_id]

    # --- exchange with a real acnportal installation ("JPL constraint set") ---
    @classmethod
    def from_acnportal(cls, cn, site: str) -> 'ChargingNetwork':
        """Reads off an ``acnportal.acnsim.ChargingNetwork`` exactly the fields the reference reads
        (env.py:133-134, 373, 451, 485-493; module docstring).  Run once where acnportal is installed::

            from acnportal.acnsim.network.sites import jpl_acn
            ChargingNetwork.from_acnportal(jpl_acn(), 'jpl').to_json('jpl_acn.json')

        The result is not provisional: it IS the reference's network."""
        station_ids = [str(s) for s in cn.station_ids]
        A = np.asarray(cn.constraint_matrix, dtype=np.float64)
        phase = np.asarray(cn._phase_angles, dtype=np.float64).reshape(-1)
        mags = np.asarray(cn.magnitudes, dtype=np.float64).reshape(-1)
        min_pilot = np.asarray(cn.min_pilot_signals, dtype=np.float64).reshape(-1)
        # env.py:373-378 tests `min_pilot_signals[i] == 6` and treats EVERY other value as the {0, 8, 16, 24, 32} A EVSE
        # (an acnportal version may report a FiniteRatesEVSE's minimum as allowable_rates[0] = 0): mirror that, and only
        # warn about values the reference's rule has never met
        if not np.isin(min_pilot, (0.0, 6.0, 8.0)).all():
            warnings.warn(f'min_pilot_signals = {sorted(set(min_pilot.tolist()))}: stations whose value is not 6 A are treated as '
                          'ClipperCreek {0, 8, 16, 24, 32} A EVSEs, as env.py:373-378 does', stacklevel=2)
        names = [str(c) for c in getattr(cn, 'constraint_index', [f'constraint {i}' for i in range(len(mags))])]
        voltages = np.asarray(getattr(cn, '_voltages', [208.0])).reshape(-1)
        return cls(site, station_ids, A, phase, mags, names, np.where(min_pilot == 6.0, EVSE_AV, EVSE_CC),
                   float(voltages[0]) if len(voltages) else 208.0, provisional=False)

    def to_json(self, path: str) -> None:
        doc = {'format': 'sustaingym_amd.ChargingNetwork/1', 'site': self.site, 'station_ids': self.station_ids,
               'constraint_matrix': self.constraint_matrix.tolist(), 'phase_angles': self.phase_angles.tolist(),
               'magnitudes': self.magnitudes.tolist(), 'constraint_names': self.constraint_names,
               'evse_kind': self.evse_kind.tolist(), 'voltage': self.voltage, 'provisional': bool(self.provisional)}
        with open(path, 'w') as f:
            json.dump(doc, f, indent=1)

    @classmethod
    def from_json(cls, path: str) -> 'ChargingNetwork':
        with open(path) as f:
            doc = json.load(f)
        if doc.get('format') != 'sustaingym_amd.ChargingNetwork/1':
            raise ValueError(f'{path}: not a sustaingym_amd ChargingNetwork file')
        return cls(doc['site'], list(doc['station_ids']), np.array(doc['constraint_matrix'], dtype=np.float64),
                   np.array(doc['phase_angles']), np.array(doc['magnitudes']), list(doc['constraint_names']),
                   np.array(doc['evse_kind'], dtype=np.uint8), float(doc.get('voltage', 208.0)),
                   provisional=bool(doc.get('provisional', False)))

    def warn_if_provisional(self) -> None:
        if self.provisional:
            warnings.warn(
                f"network {self.site!r}: this constraint set is a PROVISIONAL stand-in, not acnportal's "
                f"(the reference takes it from the un-vendored acnportal, utils.py:83-88). Action projection and "
                f"the excess_charge reward term therefore differ from the reference. Export the real one once with "
                f"ChargingNetwork.from_acnportal(...).to_json(...) and point {NETWORK_ENV_PREFIX}{self.site.upper()} "
                f"at it (INTEGRATION.md).", ProvisionalNetworkWarning, stacklevel=3)

    def a_tilde(self) -> np.ndarray:
        """Complex constraint matrix of env.py:485-486."""
        phase_factor = np.exp(1j * np.deg2rad(self.phase_angles))
        return self.constraint_matrix * phase_factor[None, :]

    def constraint_current(self, schedule: np.ndarray) -> np.ndarray:
        """``acnportal ChargingNetwork.constraint_current`` (complex aggregate currents)."""
        return self.a_tilde() @ np.asarray(schedule, dtype=np.float64)


def _current(ids: list[str], station_ids: list[str]) -> np.ndarray:
    """acnportal ``Current(ids)``: unit load on every listed EVSE."""
    v = np.zeros(len(station_ids))
    for s in ids:
        v[station_ids.index(s)] = 1.0
    return v


def caltech_acn(transformer_cap: float = 150.0, voltage: float = 208.0) -> ChargingNetwork:
    """Caltech ACN, 54 EVSEs, 8 constraints (acnportal ``sites.caltech_acn``)."""
    cc_pod = ['CA-322', 'CA-493', 'CA-496', 'CA-320', 'CA-495', 'CA-321', 'CA-323', 'CA-494']
    av_pod = ['CA-324', 'CA-325', 'CA-326', 'CA-327', 'CA-489', 'CA-490', 'CA-491', 'CA-492']
    ab_ids = [f'CA-{i}' for i in (308, 508, 303, 513, 310, 506, 316, 500, 318, 498)] + av_pod + cc_pod
    bc_ids = [f'CA-{i}' for i in (304, 512, 305, 511, 313, 503, 311, 505, 317, 499, 148, 149, 212, 213)]
    ca_ids = [f'CA-{i}' for i in (307, 509, 309, 507, 306, 510, 315, 501, 319, 497, 312, 504, 314, 502)]
    station_ids = ab_ids + bc_ids + ca_ids
    n = len(station_ids)
    phase = np.empty(n)
    kind = np.zeros(n, dtype=np.uint8)
    for i, s in enumerate(station_ids):
        phase[i] = 30.0 if s in ab_ids else (-90.0 if s in bc_ids else 150.0)
        kind[i] = EVSE_CC if s in cc_pod else EVSE_AV

    AB, BC, CA = (_current(x, station_ids) for x in (ab_ids, bc_ids, ca_ids))
    I3a, I3b, I3c = AB - CA, BC - AB, CA - BC
    I2a, I2b, I2c = 0.25 * (I3a - I3c), 0.25 * (I3b - I3a), 0.25 * (I3c - I3b)
    primary = transformer_cap * 1000 / 3 / 277
    secondary = transformer_cap * 1000 / 3 / 120
    rows = [I3a, I3b, I3c, I2a, I2b, I2c, _current(cc_pod, station_ids), _current(av_pod, station_ids)]
    mags = [secondary] * 3 + [primary] * 3 + [80.0, 80.0]
    names = ['Secondary A', 'Secondary B', 'Secondary C', 'Primary A', 'Primary B', 'Primary C',
             'CC Pod', 'AV Pod']
    return ChargingNetwork('caltech', station_ids, np.stack(rows), phase, np.array(mags), names,
                           kind, voltage)


def jpl_acn(voltage: float = 208.0) -> ChargingNetwork:
    """JPL ACN, 52 EVSEs.  **PROVISIONAL constraint set** (parity unpinned).

    Station order is the recovered ``cn.station_ids`` order.  The grouping in
    that order (14 first-floor EVSEs, then per floor 8 / 5 / 6 EVSEs) is taken to be the
    AB / BC / CA phase split of the third- and fourth-floor panels; the first floor is split
    5 / 5 / 4.  Each floor panel is modelled like the Caltech delta panel (line currents as
    phase-current differences) with a 3-phase breaker limit, plus one site transformer with
    secondary and primary limits.  All EVSEs are AeroVironment units.
    """
    f1 = ['AG-1F12', 'AG-1F14', 'AG-1F11', 'AG-1F13', 'AG-1F03', 'AG-1F06', 'AG-1F01', 'AG-1F04',
          'AG-1F02', 'AG-1F05', 'AG-1F10', 'AG-1F07', 'AG-1F09', 'AG-1F08']
    f3 = [['AG-3F16', 'AG-3F17', 'AG-3F20', 'AG-3F23', 'AG-3F25', 'AG-3F26', 'AG-3F29', 'AG-3F33'],
          ['AG-3F18', 'AG-3F21', 'AG-3F27', 'AG-3F30', 'AG-3F31'],
          ['AG-3F15', 'AG-3F19', 'AG-3F22', 'AG-3F24', 'AG-3F28', 'AG-3F32']]
    f4 = [['AG-4F35', 'AG-4F36', 'AG-4F39', 'AG-4F42', 'AG-4F44', 'AG-4F45', 'AG-4F48', 'AG-4F52'],
          ['AG-4F37', 'AG-4F40', 'AG-4F46', 'AG-4F49', 'AG-4F50'],
          ['AG-4F34', 'AG-4F38', 'AG-4F41', 'AG-4F43', 'AG-4F47', 'AG-4F51']]
    f1s = [f1[0:5], f1[5:10], f1[10:14]]
    station_ids = f1 + f3[0] + f3[1] + f3[2] + f4[0] + f4[1] + f4[2]
    n = len(station_ids)
    assert n == 52
    ab = f1s[0] + f3[0] + f4[0]
    bc = f1s[1] + f3[1] + f4[1]
    phase = np.array([30.0 if s in ab else (-90.0 if s in bc else 150.0) for s in station_ids])
    kind = np.zeros(n, dtype=np.uint8)

    rows, mags, names = [], [], []

    def delta_panel(groups, limit, label):
        a, b, c = (_current(g, station_ids) for g in groups)
        for r, nm in ((a - c, 'A'), (b - a, 'B'), (c - b, 'C')):
            rows.append(r)
            mags.append(limit)
            names.append(f'{label} {nm}')
        return a, b, c

    a1, b1, c1 = delta_panel(f1s, 200.0, 'First Floor Panel')
    a3, b3, c3 = delta_panel(f3, 225.0, 'Third Floor Panel')
    a4, b4, c4 = delta_panel(f4, 225.0, 'Fourth Floor Panel')
    AB, BC, CA = a1 + a3 + a4, b1 + b3 + b4, c1 + c3 + c4
    I3a, I3b, I3c = AB - CA, BC - AB, CA - BC
    cap = 300.0  # kVA
    for r, nm in ((I3a, 'A'), (I3b, 'B'), (I3c, 'C')):
        rows.append(r)
        mags.append(cap * 1000 / 3 / 120)
        names.append(f'Transformer Secondary {nm}')
    for r, nm in ((0.25 * (I3a - I3c), 'A'), (0.25 * (I3b - I3a), 'B'), (0.25 * (I3c - I3b), 'C')):
        rows.append(r)
        mags.append(cap * 1000 / 3 / 277)
        names.append(f'Transformer Primary {nm}')
    return ChargingNetwork('jpl', station_ids, np.stack(rows), phase, np.array(mags), names, kind,
                           voltage, provisional=True)


NETWORK_ENV_PREFIX = 'SUSTAINGYM_AMD_NETWORK_'
_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


def exported_network_path(site: str) -> str | None:
    """Where an exported acnportal network for ``site`` is looked for: ``$SUSTAINGYM_AMD_NETWORK_<SITE>``,
    then ``sustaingym_amd/data/<site>_acn.json``."""
    env = os.environ.get(NETWORK_ENV_PREFIX + site.upper())
    if env:
        return env
    packaged = os.path.join(_DATA_DIR, f'{site}_acn.json')
    return packaged if os.path.exists(packaged) else None


def site_str_to_site(site: str) -> ChargingNetwork:
    """Reference ``utils.site_str_to_site`` (utils.py:83-88).  A network exported from acnportal
    (``ChargingNetwork.from_acnportal(...).to_json``) takes precedence over the built-in restatement; its
    station order must be the one the packaged traces / GMMs are indexed by."""
    if site not in ('caltech', 'jpl'):
        raise ValueError(f"site must be 'caltech' or 'jpl', got {site!r}")
    builtin = caltech_acn() if site == 'caltech' else jpl_acn()
    path = exported_network_path(site)
    if path is None:
        return builtin
    net = ChargingNetwork.from_json(path)
    if net.station_ids != builtin.station_ids:
        raise ValueError(f'{path}: station_ids differ from the order the packaged {site} data is indexed by '
                         f'(SURVEY.md §8a); refusing to mix them')
    net.site = site
    return net


def station_groups(net: ChargingNetwork) -> tuple[np.ndarray, np.ndarray]:
    """Classes of stations with identical constraint column and phase angle.

    Every network constraint depends on the schedule only through the per-class sums, which
    is what the HIP kernels reduce over.  Returns ``(group_of_station[n], representative[G])``
    with classes numbered in order of first appearance.
    """
    keys: dict[tuple, int] = {}
    gid = np.empty(net.num_stations, dtype=np.int32)
    rep: list[int] = []
    for i in range(net.num_stations):
        key = (tuple(net.constraint_matrix[:, i].tolist()), float(net.phase_angles[i]))
        if key not in keys:
            keys[key] = len(rep)
            rep.append(i)
        gid[i] = keys[key]
    return gid, np.array(rep, dtype=np.int32)
