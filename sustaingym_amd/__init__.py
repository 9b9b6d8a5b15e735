"""MI355X-native batched step() engine for SustainGym's EV-charging environments.

Public surface (mirrors ``sustaingym.envs.evcharging``):

    EVChargingEnv, MultiAgentEVChargingEnv, DiscreteActionWrapper      (reference API)
    EVChargingVectorEnv, SB3VecEnv, RLlibVectorEnv                      (batched API + RL-library adapters)
    RealTraceGenerator, GMMsTraceGenerator, BatchedGMMTraceGenerator,
    DeviceGMMTraceGenerator, RealTraceBank                              (episode generators)
    StepEngine                                                          (C-ABI handle)
    BatteryDispatchVectorEnv                                            (config 4, synthetic: battery.py)

The HIP library is loaded lazily (``sustaingym_amd._lib.load``); importing the package does not
need a GPU, constructing an environment does.
"""
from .network import ChargingNetwork, caltech_acn, jpl_acn, site_str_to_site  # noqa: F401
from .event_generation import (AbstractTraceGenerator, BatchedGMMTraceGenerator,  # noqa: F401
                               DeviceGMMTraceGenerator, EventTable, GMMsTraceGenerator, MOERLoader,
                               RealTraceBank, RealTraceGenerator)


def __getattr__(name):
    if name == 'StepEngine':
        from .engine import StepEngine
        return StepEngine
    if name == 'BatteryDispatchVectorEnv':
        from .battery import BatteryDispatchVectorEnv
        return BatteryDispatchVectorEnv
    if name in ('EVChargingEnv', 'MultiAgentEVChargingEnv', 'DiscreteActionWrapper',
                'EVChargingVectorEnv', 'SB3VecEnv', 'RLlibVectorEnv', 'MultiAgentEVChargingVectorEnv'):
        from . import envs
        return getattr(envs, name)
    raise AttributeError(name)


def _register_with_gymnasium() -> None:
    """sustaingym/__init__.py:1-7 registers 'sustaingym/EVCharging-v0'; with gymnasium installed the
    same id (and 'sustaingym_amd/EVCharging-v0') resolves to this package's EVChargingEnv."""
    try:
        from gymnasium.envs.registration import register, registry
    except Exception:       # gymnasium absent (this build image): nothing to register with
        return
    for env_id in ('sustaingym/EVCharging-v0', 'sustaingym_amd/EVCharging-v0'):
        if env_id not in registry:
            register(id=env_id, entry_point='sustaingym_amd.envs:EVChargingEnv', nondeterministic=False)


_register_with_gymnasium()

__version__ = '0.1.0'
