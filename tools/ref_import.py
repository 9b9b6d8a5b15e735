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


def install_step_stubs() -> None:
    """Container-only stubs that let the PYTHON HALF of the step import unmodified:
    ``sustaingym.envs.evcharging.env`` (EVChargingEnv: class constants env.py:99-114, ``__init__`` :116-176,
    ``_to_schedule`` :340-379, ``_get_observation`` :381-394, ``_get_reward`` :431-464),
    ``sustaingym.envs.wrappers`` (DiscreteActionWrapper.action :43-45) and
    ``sustaingym.envs.evcharging.multiagent_env`` (``_create_dict_from_obs_agg`` :102-148).

    What is stubbed are base classes and containers, NOT arithmetic:
    * ``gymnasium.Env`` / ``ActionWrapper`` / ``pettingzoo.ParallelEnv``: empty subscriptable base classes
      (``ActionWrapper.__init__`` stores ``env``; ``ParallelEnv.num_agents`` = ``len(self.agents)``);
    * ``gymnasium.spaces.Box / Dict / Discrete / MultiDiscrete``: attribute bags; ``Dict`` keeps its sub-spaces in SORTED
      key order and ``spaces.flatten`` concatenates the raveled values in that order — gymnasium 0.28's behaviour for a
      plain-dict Dict space, restated from memory ([MEM]: gymnasium is not in the image).  That ordering is the one thing
      here that is not a pure container; it moves no number;
    * ``cvxpy``: importable, every attribute raises if it is touched (the projection cannot run here);
    * the simulator / interface / network objects the env methods read are supplied by the caller
      (tests/golden/make_step_unit_golden.py) as attribute bags whose values are FIXTURE INPUTS."""
    install_stubs()
    if 'gymnasium.spaces' in sys.modules:
        return
    import collections
    import typing

    import numpy as np

    class _Generic:
        def __class_getitem__(cls, item):
            return cls

    gym = sys.modules['gymnasium']

    class Env(_Generic):
        def reset(self, *, seed=None, options=None):
            return None

    class ActionWrapper(_Generic):
        def __init__(self, env):
            self.env = env

    spaces = types.ModuleType('gymnasium.spaces')

    class Box(_Bag):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            super().__init__(low=low, high=high, shape=tuple(shape) if shape is not None else (), dtype=np.dtype(dtype))

    class Dict(_Bag):
        def __init__(self, spaces_):
            super().__init__(spaces=collections.OrderedDict(sorted(spaces_.items())))

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()

    class Discrete(_Bag):
        def __init__(self, n):
            super().__init__(n=int(n), shape=(), dtype=np.dtype(np.int64))

    class MultiDiscrete(_Bag):
        def __init__(self, nvec):
            super().__init__(nvec=np.asarray(nvec), shape=np.asarray(nvec).shape, dtype=np.dtype(np.int64))

    def flatten(space, x):
        if isinstance(space, Dict):
            return np.concatenate([np.asarray(x[k], dtype=space[k].dtype).ravel() for k in space.keys()])
        return np.asarray(x, dtype=space.dtype).ravel()

    def flatten_space(space):
        if isinstance(space, Dict):
            size = sum(int(np.prod(s.shape)) for s in space.spaces.values())
            return Box(None, None, shape=(size,), dtype=np.float32)
        return Box(space.low, space.high, shape=(int(np.prod(space.shape)),), dtype=space.dtype)

    spaces.Box, spaces.Dict, spaces.Discrete, spaces.MultiDiscrete = Box, Dict, Discrete, MultiDiscrete
    spaces.Space = _Bag
    spaces.flatten, spaces.flatten_space = flatten, flatten_space
    core = types.ModuleType('gymnasium.core')
    core.ObsType = typing.TypeVar('ObsType')
    gym.Env, gym.ActionWrapper, gym.spaces, gym.core = Env, ActionWrapper, spaces, core
    sys.modules.update({'gymnasium.spaces': spaces, 'gymnasium.core': core})

    pz = types.ModuleType('pettingzoo')

    class ParallelEnv(_Generic):
        @property
        def num_agents(self):
            return len(self.agents)

    pz.ParallelEnv = ParallelEnv
    sys.modules['pettingzoo'] = pz

    class _Untouchable(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith('__'):
                raise AttributeError(name)
            raise RuntimeError(f'cvxpy.{name}: cvxpy is not in this image; the projection cannot run here')

    sys.modules['cvxpy'] = _Untouchable('cvxpy')


def reference_step_modules():
    """Returns the reference modules (env, wrappers, multiagent_env), imported unmodified from /root/reference."""
    install_step_stubs()
    import importlib
    env = importlib.import_module('sustaingym.envs.evcharging.env')
    wr = importlib.import_module('sustaingym.envs.wrappers')
    ma = importlib.import_module('sustaingym.envs.evcharging.multiagent_env')
    return env, wr, ma


def reference_generators():
    """Returns the reference modules (event_generation, utils, load_moer)."""
    install_stubs()
    import importlib
    eg = importlib.import_module('sustaingym.envs.evcharging.event_generation')
    ut = importlib.import_module('sustaingym.envs.evcharging.utils')
    lm = importlib.import_module('sustaingym.data.load_moer')
    return eg, ut, lm
