"""Container-only importer of the reference's *data side* (never shipped to the GPU box).

The reference's hot path cannot be imported here (gymnasium, acnportal, cvxpy, mosek,
pettingzoo are absent — SURVEY.md §8c), but its episode-data code can, once the missing
*containers* are stubbed: ``gymnasium.envs.registration.register`` (no-op) and the
``acnportal.acnsim`` constructors (plain attribute bags; no arithmetic) plus the two site
factories, which only need ``station_ids``.  With those stubs

    sustaingym.data.load_moer.MOERLoader
    sustaingym.envs.evcharging.event_generation.RealTraceGenerator / GMMsTraceGenerator

import and run unmodified from /root/reference.  Used by tests/golden/make_golden.py and
tools/build_data.py to generate fixtures / packaged derived data.
"""
from __future__ import annotations

import os
import sys
import types
import warnings

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Bag:
    def __init__(self, *args, **kwargs):
        self.args = args
        self.__dict__.update(kwargs)


def install_stubs() -> None:
    if 'sustaingym' in sys.modules:
        return
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from sustaingym_amd.network import caltech_acn, jpl_acn

    gym = types.ModuleType('gymnasium')
    gym_envs = types.ModuleType('gymnasium.envs')
    gym_reg = types.ModuleType('gymnasium.envs.registration')
    gym_reg.register = lambda *a, **k: None
    gym.envs = gym_envs
    gym_envs.registration = gym_reg
    sys.modules.update({'gymnasium': gym, 'gymnasium.envs': gym_envs,
                        'gymnasium.envs.registration': gym_reg})

    acnportal = types.ModuleType('acnportal')
    acnsim = types.ModuleType('acnportal.acnsim')
    acndata = types.ModuleType('acnportal.acndata')

    class EV(_Bag):
        def __init__(self, arrival, departure, requested_energy, station_id, session_id, battery,
                     estimated_departure=None):
            super().__init__(arrival=arrival, departure=departure, requested_energy=requested_energy,
                             station_id=station_id, session_id=session_id, battery=battery,
                             estimated_departure=estimated_departure)

    class PluginEvent(_Bag):
        def __init__(self, timestamp, ev):
            super().__init__(timestamp=timestamp, ev=ev)

    class RecomputeEvent(_Bag):
        def __init__(self, timestamp):
            super().__init__(timestamp=timestamp)

    class EventQueue(_Bag):
        def __init__(self, events=None):
            super().__init__(events=list(events or []))

    acnsim.EV = EV
    acnsim.Linear2StageBattery = type('Linear2StageBattery', (_Bag,), {})
    acnsim.PluginEvent = PluginEvent
    acnsim.RecomputeEvent = RecomputeEvent
    acnsim.EventQueue = EventQueue
    acnsim.ChargingNetwork = _Bag
    network = types.ModuleType('acnportal.acnsim.network')
    sites = types.ModuleType('acnportal.acnsim.network.sites')
    sites.caltech_acn = lambda *a, **k: _Bag(station_ids=list(caltech_acn().station_ids))
    sites.jpl_acn = lambda *a, **k: _Bag(station_ids=list(jpl_acn().station_ids))
    network.sites = sites
    acnsim.network = network
    acnportal.acnsim = acnsim
    acnportal.acndata = acndata
    sys.modules.update({'acnportal': acnportal, 'acnportal.acnsim': acnsim,
                        'acnportal.acndata': acndata, 'acnportal.acnsim.network': network,
                        'acnportal.acnsim.network.sites': sites})

    # sustaingym/envs/evcharging/__init__.py imports a module that does not exist in the
    # snapshot; enter the package through synthetic package objects.
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name, rel in (('sustaingym', 'sustaingym'), ('sustaingym.envs', 'sustaingym/envs'),
                      ('sustaingym.envs.evcharging', 'sustaingym/envs/evcharging')):
        import importlib.machinery
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF, rel)]
        spec = importlib.machinery.ModuleSpec(name, None, is_package=True)
        spec.submodule_search_locations = [os.path.join(REF, rel)]
        mod.__spec__ = spec               # importlib.resources.files('sustaingym') needs a spec
        sys.modules[name] = mod
    warnings.filterwarnings('ignore')
    # sustaingym.data.utils.read_bytes goes through importlib.resources, which needs a real
    # package loader; point it at the files directly (a file reader, no arithmetic).
    import importlib
    du = importlib.import_module('sustaingym.data.utils')

    def read_bytes(path: str) -> bytes:
        with open(os.path.join(REF, 'sustaingym', path), 'rb') as f:
            return f.read()
    du.read_bytes = read_bytes


def reference_generators():
    """Returns the reference modules (event_generation, utils, load_moer)."""
    install_stubs()
    import importlib
    eg = importlib.import_module('sustaingym.envs.evcharging.event_generation')
    ut = importlib.import_module('sustaingym.envs.evcharging.utils')
    lm = importlib.import_module('sustaingym.data.load_moer')
    return eg, ut, lm
