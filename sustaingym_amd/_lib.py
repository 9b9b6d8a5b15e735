"""ctypes loader of the C-ABI shared library ``libevcharge_hip.so`` (include/evcharge.h).

The library is the product: hand-written gfx950 HIP kernels behind a plain-C interface.  There
is deliberately no fallback of any kind here — if the library has not been built
(``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C sustaingym_amd/csrc``) or
no MI355X is visible, loading / ``evc_create`` fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SUSTAINGYM_AMD_LIB', os.path.join(_PKG, 'libevcharge_hip.so'))
HEADER_PATH = os.path.join(os.path.dirname(_PKG), 'include', 'evcharge.h')

ABI_VERSION = 7
MAX_STATIONS, MAX_CONSTRAINTS, MAX_GROUPS, MAX_SESSIONS = 64, 32, 16, 256
MOER_ROWS, MOER_COLS, EPISODE_STEPS = 289, 37, 288

FLAG_PROJECT_ACTION = 1 << 0
FLAG_AUTORESET = 1 << 1
FLAG_BATTERY_STEPWISE = 1 << 2    # acnportal Linear2StageBattery(charge_calculation='stepwise'), the legacy model
ACTION_F32, ACTION_DISCRETE, ACTION_GREEDY, ACTION_RANDOM = 0, 1, 2, 3

STATUS_OCCUPIED = 1 << 0
STATUS_PROJ_NOCONV = 1 << 1
STATUS_STEP_AFTER_DONE = 1 << 2
STATUS_ACTION_CLAMPED = 1 << 3

SESSION_DTYPE = np.dtype([('arrival', '<i2'), ('departure', '<i2'),
                          ('est_departure', '<i2'), ('station', '<i2')])


class NetworkDesc(C.Structure):
    _fields_ = [('n_stations', C.c_int32), ('n_constraints', C.c_int32),
                ('constraint_matrix', C.c_void_p), ('phase_angles_deg', C.c_void_p),
                ('magnitudes', C.c_void_p), ('evse_kind', C.c_void_p)]


class GmmDesc(C.Structure):
    _fields_ = [('n_components', C.c_int32), ('n_counts', C.c_int32), ('num_days', C.c_int32),
                ('reserved', C.c_int32), ('cum_weights', C.c_void_p), ('means', C.c_void_p),
                ('chol', C.c_void_p), ('daily_counts', C.c_void_p), ('station_usage', C.c_void_p),
                ('requested_energy_cap', C.c_double)]


class StepOut(C.Structure):
    _fields_ = [('obs', C.c_void_p), ('reward', C.c_void_p), ('terminated', C.c_void_p),
                ('breakdown', C.c_void_p), ('final_obs', C.c_void_p), ('pilots', C.c_void_p),
                ('rates', C.c_void_p), ('projected', C.c_void_p), ('returns', C.c_void_p)]


class EngineLibraryError(RuntimeError):
    pass


# Every symbol include/evcharge.h declares: name -> (restype, argtypes)
_vp, _i32, _u32 = C.c_void_p, C.c_int32, C.c_uint32
SIGNATURES = {
    'evc_create': (_i32, [C.POINTER(NetworkDesc), _i32, _i32, _u32, _i32, _i32, _i32, _i32, C.POINTER(_vp)]),
    'evc_destroy': (None, [_vp]),
    'evc_last_error': (C.c_char_p, []),
    'evc_abi_version': (_i32, []),
    'evc_set_stream': (_i32, [_vp, _vp]),
    'evc_synchronize': (_i32, [_vp]),
    'evc_set_pipeline': (_i32, [_vp, _i32]),
    'evc_join': (_i32, [_vp]),
    'evc_pipeline_half': (_i32, [_vp, _i32, C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_i32)]),
    'evc_obs_dim': (_i32, [_vp]),
    'evc_num_envs': (_i32, [_vp]),
    'evc_num_stations': (_i32, [_vp]),
    'evc_num_groups': (_i32, [_vp]),
    'evc_upload_moer': (_i32, [_vp, _i32, _i32, _vp]),
    'evc_upload_episodes': (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    'evc_set_autoreset_stride': (_i32, [_vp, _i32]),
    'evc_upload_gmm': (_i32, [_vp, C.POINTER(GmmDesc)]),
    'evc_generate_episodes': (_i32, [_vp, _i32, _i32, C.c_uint64, C.c_uint64]),
    'evc_download_episodes': (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'evc_reset': (_i32, [_vp, _vp, _i32, _vp, _vp]),
    'evc_step': (_i32, [_vp, _vp, _i32, _i32, C.POINTER(StepOut)]),
    'evc_rollout': (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, C.POINTER(StepOut)]),
    'evc_set_policy_seed': (_i32, [_vp, C.c_uint64, _u32]),
    'evc_last_rollout_waves': (_i32, [_vp, C.POINTER(_i32)]),
    'evc_fill_random_actions': (_i32, [_vp, _i32, _vp]),
    'evc_gather_agent_obs': (_i32, [_vp, _vp, _vp, _vp]),
    'evc_host_register': (_i32, [_vp, C.c_size_t]),
    'evc_host_unregister': (_i32, [_vp]),
    'evc_reset_host': (_i32, [_vp, _vp, _i32, _vp, _vp]),
    'evc_step_host': (_i32, [_vp, _vp, _i32, _i32, C.POINTER(StepOut)]),
    'evc_get_env_scalars': (_i32, [_vp, _vp]),
    'evc_set_env_scalars': (_i32, [_vp, _vp]),
    'evc_get_station_state': (_i32, [_vp, _vp, _vp, _vp]),
    'evc_set_station_state': (_i32, [_vp, _vp, _vp, _vp]),
    'evc_get_entry_rank': (_i32, [_vp, _vp]),
    'evc_set_station_state_ranked': (_i32, [_vp, _vp, _vp, _vp, _vp]),
    'evc_get_breakdown': (_i32, [_vp, _vp]),
    'evc_set_breakdown': (_i32, [_vp, _vp]),
    'evc_clear_status': (_i32, [_vp]),
    'evc_read_metrics': (_i32, [_vp, _vp]),
    'evc_set_tie_grid': (_i32, [_vp, _i32]),
    'evc_last_slow_count': (_i32, [_vp, C.POINTER(_i32)]),
    'evc_enable_timing': (_i32, [_vp, _i32]),
    'evc_last_step_ms': (_i32, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    'evc_last_half_ms': (_i32, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    'evc_pipelined_steps': (_i32, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
}

class BatConfig(C.Structure):
    _fields_ = [('num_envs', C.c_int32), ('forecast_steps', C.c_int32), ('bank_slots', C.c_int32),
                ('device', C.c_int32), ('capacity_mwh', C.c_double), ('max_power_mw', C.c_double),
                ('eta_charge', C.c_double), ('eta_discharge', C.c_double), ('init_energy_mwh', C.c_double),
                ('co2_price_per_kg', C.c_double)]


# Every symbol include/battery_dispatch.h declares
BAT_SIGNATURES = {
    'bat_create': (_i32, [C.POINTER(BatConfig), C.POINTER(_vp)]),
    'bat_destroy': (None, [_vp]),
    'bat_last_error': (C.c_char_p, []),
    'bat_obs_dim': (_i32, [_vp]),
    'bat_set_stream': (_i32, [_vp, _vp]),
    'bat_upload_traces': (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    'bat_reset': (_i32, [_vp, _vp, _vp]),
    'bat_step': (_i32, [_vp, _vp, _vp, _vp, _vp]),
    'bat_rollout': (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'bat_rollout_pitched': (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    'bat_reset_host': (_i32, [_vp, _vp, _vp]),
    'bat_step_host': (_i32, [_vp, _vp, _vp, _vp, _vp]),
    'bat_get_state': (_i32, [_vp, _vp, _vp]),
    'bat_read_metrics': (_i32, [_vp, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Loads the HIP engine library.  Raises ``EngineLibraryError`` if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineLibraryError(
            f'{LIB_PATH} not found: the gfx950 HIP engine has not been built. Run '
            '`python -c "import __graft_entry__ as g; g.build()"` (or `make -C sustaingym_amd/csrc`). '
            'There is no CPU fallback.')
    # PyTorch-ROCm bundles its own libamdhip64; if it is going to be used in this process it
    # must be the one HIP runtime both sides share, so let it load first.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional plumbing: the host-buffer entry points do not need it
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # e.g. libamdhip64 missing
        raise EngineLibraryError(f'cannot load {LIB_PATH}: {exc}') from exc
    for name, (res, args) in list(SIGNATURES.items()) + list(BAT_SIGNATURES.items()):
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise EngineLibraryError(f'{LIB_PATH} does not export {name}') from exc
        fn.restype = res
        fn.argtypes = args
    if lib.evc_abi_version() != ABI_VERSION:
        raise EngineLibraryError(f'ABI version mismatch: library {lib.evc_abi_version()} != {ABI_VERSION}')
    _lib = lib
    return lib


def check(rc: int, what: str = '') -> None:
    if rc != 0:
        msg = load().evc_last_error().decode('utf-8', 'replace')
        raise EngineLibraryError(f'{what} failed with code {rc}: {msg}')
