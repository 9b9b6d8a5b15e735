"""Python handle on the C-ABI step engine (include/evcharge.h).

``StepEngine`` is plumbing only: it marshals numpy / torch buffers into the C calls.  Device
buffers are torch CUDA(HIP) tensors (PyTorch is used for device memory and streams, nothing
else); numpy inputs go through the ``*_host`` entry points.
"""
from __future__ import annotations

import ctypes as C
from typing import Any

import numpy as np

from . import _lib
from ._lib import SESSION_DTYPE, NetworkDesc, StepOut, check
from .network import ChargingNetwork


def _np_ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class StepEngine:
    """Batched EVChargingEnv.step()/reset() engine for ``num_envs`` environments on one GPU.

    Replaces, for a batch, ``EVChargingEnv.__init__/reset/step/close`` of the reference
    (sustaingym/envs/evcharging/env.py:116-176, 293-338, 229-291, 466-470).

    ``charge_calculation`` is the argument of that name of acnportal's ``Linear2StageBattery``: the
    reference does not pass it (event_generation.py:173-176) and so runs acnportal's default
    ``'continuous'`` model, the default here; ``'stepwise'`` selects acnportal's legacy model.
    """

    def __init__(self, network: ChargingNetwork, num_envs: int, moer_forecast_steps: int = 36,
                 project_action: bool = True, autoreset: bool = False, device: int = 0,
                 bank_slots: int | None = None, max_sessions: int = 128, moer_days: int = 1,
                 debug_outputs: bool = False, charge_calculation: str = 'continuous'):
        if charge_calculation not in ('continuous', 'stepwise'):       # acnportal raises ValueError too
            raise ValueError("charge_calculation must be 'continuous' (acnportal's default) or 'stepwise'")
        self.charge_calculation = charge_calculation
        self.lib = _lib.load()
        self.network = network
        if hasattr(network, 'warn_if_provisional'):
            network.warn_if_provisional()       # e.g. the built-in JPL constraint set (network.py)
        self.N = int(num_envs)
        self.n = network.num_stations
        self.k = int(moer_forecast_steps)
        self.F = 2 * self.n + self.k + 2
        self.project_action = bool(project_action)
        self.autoreset = bool(autoreset)
        self.device = int(device)
        self.bank_slots = int(bank_slots if bank_slots is not None else num_envs)
        self.max_sessions = int(max_sessions)
        self.moer_days = int(moer_days)
        self.debug_outputs = bool(debug_outputs)
        self.autoreset_stride = 1
        self._A = np.ascontiguousarray(network.constraint_matrix, dtype=np.float64)
        self._ph = np.ascontiguousarray(network.phase_angles, dtype=np.float64)
        self._mag = np.ascontiguousarray(network.magnitudes, dtype=np.float64)
        self._kind = np.ascontiguousarray(network.evse_kind, dtype=np.uint8)
        desc = NetworkDesc(self.n, self._A.shape[0], _np_ptr(self._A), _np_ptr(self._ph),
                           _np_ptr(self._mag), _np_ptr(self._kind))
        flags = (_lib.FLAG_PROJECT_ACTION if project_action else 0) | \
                (_lib.FLAG_AUTORESET if autoreset else 0) | \
                (_lib.FLAG_BATTERY_STEPWISE if charge_calculation == 'stepwise' else 0)
        handle = C.c_void_p()
        check(self.lib.evc_create(C.byref(desc), self.N, self.k, flags, self.device,
                                  self.bank_slots, self.max_sessions, self.moer_days,
                                  C.byref(handle)), 'evc_create')
        self.handle = handle
        self._pipeline = 1
        self._dev_out: dict[str, Any] | None = None
        self._host_out = None
        self._registered: list[np.ndarray] = []
        self._returns = None

    # ------------------------------------------------------------------ lifecycle
    def close(self) -> None:
        if getattr(self, 'handle', None):
            for arr in getattr(self, '_registered', []):      # arrays stay valid (pageable again)
                self.lib.evc_host_unregister(C.c_void_p(arr.ctypes.data))
            self._registered = []
            self.lib.evc_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_groups(self) -> int:
        return self.lib.evc_num_groups(self.handle)

    # ------------------------------------------------------------------ episode data
    def upload_moer(self, moer: np.ndarray, first_day: int = 0) -> None:
        """moer: float64 [D, 289, 37] (``get_moer()`` matrices, event_generation.py:209-218)."""
        moer = np.ascontiguousarray(moer, dtype=np.float64)
        if moer.ndim == 2:
            moer = moer[None]
        assert moer.shape[1:] == (_lib.MOER_ROWS, _lib.MOER_COLS), moer.shape
        check(self.lib.evc_upload_moer(self.handle, first_day, moer.shape[0], _np_ptr(moer)),
              'evc_upload_moer')

    def upload_episodes(self, n_sessions, sessions, requested, moer_day, first_slot: int = 0) -> None:
        """Episodes = event tables (arrival-sorted sessions) + MOER day slot per episode."""
        n_sessions = np.ascontiguousarray(n_sessions, dtype=np.int32)
        sessions = np.ascontiguousarray(sessions, dtype=SESSION_DTYPE)
        requested = np.ascontiguousarray(requested, dtype=np.float64)
        moer_day = np.ascontiguousarray(moer_day, dtype=np.int32)
        if sessions.ndim == 1:
            sessions, requested = sessions[None], requested[None]
        count, stride = sessions.shape
        assert requested.shape == (count, stride) and n_sessions.shape == (count,) == moer_day.shape
        check(self.lib.evc_upload_episodes(self.handle, first_slot, count, stride, _np_ptr(n_sessions),
                                           _np_ptr(sessions), _np_ptr(requested), _np_ptr(moer_day)),
              'evc_upload_episodes')

    def upload_gmm(self, tables: dict) -> None:
        """Model of the on-device episode generator (``event_generation.gmm_device_tables``)."""
        keep = {
            'cum_weights': np.ascontiguousarray(tables['cum_weights'], dtype=np.float64),
            'means': np.ascontiguousarray(tables['means'], dtype=np.float64),
            'chol': np.ascontiguousarray(tables['chol'], dtype=np.float64),
            'daily_counts': np.ascontiguousarray(tables['daily_counts'], dtype=np.int32),
            'station_usage': np.ascontiguousarray(tables['station_usage'], dtype=np.uint32),
        }
        K = len(keep['cum_weights'])
        assert keep['means'].shape == (K, 4) and keep['chol'].shape == (K, 4, 4)
        assert len(keep['station_usage']) == self.n
        desc = _lib.GmmDesc(K, len(keep['daily_counts']), int(tables['num_days']), 0,
                            keep['cum_weights'].ctypes.data, keep['means'].ctypes.data, keep['chol'].ctypes.data,
                            keep['daily_counts'].ctypes.data, keep['station_usage'].ctypes.data,
                            float(tables['requested_energy_cap']))
        check(self.lib.evc_upload_gmm(self.handle, C.byref(desc)), 'evc_upload_gmm')

    def generate_episodes(self, first_slot: int, count: int, seed: int, first_episode: int) -> None:
        """Fills bank slots ``[first_slot, first_slot+count)`` on the GPU (asynchronous on the
        current stream): episode ``first_episode + i`` of the Philox stream keyed by ``seed``."""
        import sys
        if 'torch' in sys.modules and sys.modules['torch'].cuda.is_available():
            self._bind_stream()                  # otherwise the engine's own (host-path) stream
        check(self.lib.evc_generate_episodes(self.handle, int(first_slot), int(count),
                                             C.c_uint64(int(seed) & (2 ** 64 - 1)),
                                             C.c_uint64(int(first_episode))), 'evc_generate_episodes')

    def download_episodes(self, first_slot: int, count: int, tables: bool = True):
        """Bank slots back on the host: ``(n_sessions, sessions, requested, moer_day, max_profit)``
        (``sessions``/``requested`` are ``None`` when ``tables`` is false)."""
        ns = np.zeros(count, np.int32)
        day = np.zeros(count, np.int32)
        mp = np.zeros(count, np.float64)
        sess = np.zeros((count, self.max_sessions), dtype=SESSION_DTYPE) if tables else None
        req = np.zeros((count, self.max_sessions), dtype=np.float64) if tables else None
        check(self.lib.evc_download_episodes(self.handle, int(first_slot), count, self.max_sessions,
                                             _np_ptr(ns), None if sess is None else _np_ptr(sess),
                                             None if req is None else _np_ptr(req), _np_ptr(day), _np_ptr(mp)),
              'evc_download_episodes')
        return ns, sess, req, day, mp

    def set_autoreset_stride(self, stride: int) -> None:
        check(self.lib.evc_set_autoreset_stride(self.handle, int(stride)), 'evc_set_autoreset_stride')
        self.autoreset_stride = int(stride) % self.bank_slots

    # ------------------------------------------------------------------ device buffers
    def _torch(self):
        import torch
        if not torch.cuda.is_available():
            raise _lib.EngineLibraryError('no HIP device visible to torch')
        return torch

    def _bind_stream(self) -> None:
        torch = self._torch()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.evc_set_stream(self.handle, C.c_void_p(stream)), 'evc_set_stream')

    def device_outputs(self) -> dict[str, Any]:
        if self._dev_out is None:
            torch = self._torch()
            dev = torch.device('cuda', self.device)
            N, n, F = self.N, self.n, self.F
            out = {
                'obs': torch.zeros((N, F), dtype=torch.float32, device=dev),
                'reward': torch.zeros((N,), dtype=torch.float64, device=dev),
                'terminated': torch.zeros((N,), dtype=torch.uint8, device=dev),
                'breakdown': torch.zeros((N, 3), dtype=torch.float64, device=dev),
                'final_obs': torch.zeros((N, F), dtype=torch.float32, device=dev),
            }
            if self.debug_outputs:
                for key in ('pilots', 'rates', 'projected'):
                    out[key] = torch.zeros((N, n), dtype=torch.float64, device=dev)
            self._dev_out = out
        return self._dev_out

    def _step_out_struct(self, bufs: dict[str, Any], ptr) -> StepOut:
        so = StepOut()
        for name, _ in StepOut._fields_:
            setattr(so, name, ptr(bufs[name]) if name in bufs else None)
        return so

    # ------------------------------------------------------------------ hot path
    def reset(self, env_ids=None, slots=None, host: bool = False):
        """EVChargingEnv.reset for a set of environments; returns the [N, F] observation buffer."""
        ids = None if env_ids is None else np.ascontiguousarray(env_ids, dtype=np.int32)
        sl = None if slots is None else np.ascontiguousarray(slots, dtype=np.int32)
        count = self.N if ids is None else len(ids)
        if sl is not None:
            assert len(sl) == count
        if host:
            out = self._host_buffers()
            check(self.lib.evc_reset_host(self.handle, _np_ptr(ids), count, _np_ptr(sl),
                                          _np_ptr(out['obs'])), 'evc_reset_host')
            return out['obs']
        self._bind_stream()
        out = self.device_outputs()
        check(self.lib.evc_reset(self.handle, _np_ptr(ids), count, _np_ptr(sl),
                                 C.c_void_p(out['obs'].data_ptr())), 'evc_reset')
        return out['obs']

    _POLICY_KINDS = {'greedy': _lib.ACTION_GREEDY, 'random': _lib.ACTION_RANDOM}

    def step(self, actions=None, bins: int = 0, policy: str | None = None):
        """EVChargingEnv.step for all environments.

        ``actions``: torch CUDA tensor ``[N, n]`` (float32, or int64 with ``bins``) -> returns the
        dict of device output tensors (asynchronous on the current torch stream); or a numpy
        array -> synchronous host path returning numpy arrays.
        ``policy='greedy'`` / ``'random'`` (``actions=None``): the device-resident baselines (baselines.py:22-51) act instead —
        no action buffer is read; device outputs (the compact streaming kernels apply the greedy rule themselves).
        """
        if policy is not None:
            assert actions is None, 'policy= and actions are exclusive'
            kind, ptr = self._POLICY_KINDS[policy], None
        elif isinstance(actions, np.ndarray):
            return self._step_host(actions, bins)
        else:
            torch = self._torch()
            assert actions.is_cuda and actions.shape == (self.N, self.n) and actions.is_contiguous()
            if bins > 0:
                assert actions.dtype == torch.int64
                kind = _lib.ACTION_DISCRETE
            else:
                assert actions.dtype == torch.float32
                kind = _lib.ACTION_F32
            ptr = C.c_void_p(actions.data_ptr())
        self._bind_stream()
        out = self.device_outputs()
        so = self._step_out_struct(out, lambda t: C.c_void_p(t.data_ptr()))
        check(self.lib.evc_step(self.handle, ptr, kind, bins, C.byref(so)), 'evc_step')
        if self._pipeline == 2:          # the returned tensors are consumed on the torch stream: order it after both halves
            check(self.lib.evc_join(self.handle), 'evc_join')
        return out

    def make_stepper(self, bins: int = 0, policy: str | None = None):
        """Lean per-step callable for rollout loops: binds the current torch stream once and
        returns ``(step(ptr: int) -> None, outputs)`` where ``ptr`` is the device address of a
        contiguous ``[N, n]`` action tensor (float32, or int64 when ``bins > 0``); with ``policy`` ('greedy' / 'random': the
        device-resident baselines) ``ptr`` is ignored (pass ``None``)."""
        self._bind_stream()
        out = self.device_outputs()
        so = self._step_out_struct(out, lambda t: C.c_void_p(t.data_ptr()))
        ref = C.byref(so)
        fn, handle = self.lib.evc_step, self.handle
        kind = _lib.ACTION_DISCRETE if bins > 0 else _lib.ACTION_F32
        if policy is not None:
            kind = self._POLICY_KINDS[policy]

            def step_policy(ptr=None, _keep=so) -> None:
                rc = fn(handle, None, kind, bins, ref)
                if rc:
                    check(rc, 'evc_step')
            return step_policy, out

        def step(ptr: int, _keep=so) -> None:
            rc = fn(handle, ptr, kind, bins, ref)
            if rc:
                check(rc, 'evc_step')
        return step, out

    def rollout(self, actions=None, steps: int = 288, policy: str | None = None, bins: int = 0,
                accumulate_returns: bool = True):
        """``steps`` consecutive steps without returning to Python (``evc_rollout``).

        ``policy='greedy'``: device-resident GreedyAlgorithm (baselines.py:22-35), no action buffer.
        ``policy='random'``: device-resident RandomAlgorithm (baselines.py:38-51) on the counter-based
        stream of :meth:`set_policy_seed`; ``bins >= 2`` draws DiscreteActionWrapper levels.
        Otherwise ``actions`` is a contiguous CUDA tensor ``[R, N, n]`` (float32, or int64 with
        ``bins``) used as a ring.  Returns the device outputs of the last step plus ``'returns'``
        (sum of rewards per environment over the rollout)."""
        torch = self._torch()
        self._bind_stream()
        out = dict(self.device_outputs())
        if accumulate_returns:
            if self._returns is None:
                self._returns = torch.zeros((self.N,), dtype=torch.float64,
                                            device=torch.device('cuda', self.device))
            self._returns.zero_()
            out['returns'] = self._returns
        so = self._step_out_struct(out, lambda t: C.c_void_p(t.data_ptr()))
        if policy == 'greedy':
            kind, ptr, ring = _lib.ACTION_GREEDY, None, 1
        elif policy == 'random':
            kind, ptr, ring = _lib.ACTION_RANDOM, None, 1
        elif policy is not None:
            raise ValueError(f'unknown device policy {policy!r}')
        else:
            assert actions is not None and actions.is_cuda and actions.is_contiguous()
            assert actions.shape[1:] == (self.N, self.n)
            kind = _lib.ACTION_DISCRETE if bins > 0 else _lib.ACTION_F32
            ptr, ring = C.c_void_p(actions.data_ptr()), actions.shape[0]
        check(self.lib.evc_rollout(self.handle, ptr, kind, bins, int(steps), int(ring), C.byref(so)),
              'evc_rollout')
        if self._pipeline == 2:
            check(self.lib.evc_join(self.handle), 'evc_join')
        return out

    def last_rollout_waves(self) -> int:
        """Register budget (wavefronts per SIMD) of the last fused rollout launch: the engine times its launches and keeps the
        faster of the two builds of the projecting kernels (``evc_last_rollout_waves``)."""
        w = C.c_int32()
        check(self.lib.evc_last_rollout_waves(self.handle, C.byref(w)), 'evc_last_rollout_waves')
        return w.value

    def set_policy_seed(self, seed: int, env_id_base: int = 0) -> None:
        """Seed of the device-resident random policy; ``env_id_base`` = global id of environment 0."""
        check(self.lib.evc_set_policy_seed(self.handle, C.c_uint64(int(seed) & (2 ** 64 - 1)), int(env_id_base)),
              'evc_set_policy_seed')

    def fill_random_actions(self, out=None, bins: int = 0):
        """The actions ``policy='random'`` would apply now: CUDA float32 ``[N, n]``."""
        torch = self._torch()
        self._bind_stream()
        if out is None:
            out = torch.empty((self.N, self.n), dtype=torch.float32, device=torch.device('cuda', self.device))
        check(self.lib.evc_fill_random_actions(self.handle, int(bins), C.c_void_p(out.data_ptr())),
              'evc_fill_random_actions')
        return out

    def step_policy(self, policy: str, bins: int = 0):
        """One step of a device-resident policy (``'greedy'`` / ``'random'``), host buffers."""
        kind = {'greedy': _lib.ACTION_GREEDY, 'random': _lib.ACTION_RANDOM}[policy]
        out = self._host_buffers()
        so = self._step_out_struct(out, _np_ptr)
        check(self.lib.evc_step_host(self.handle, None, kind, int(bins), C.byref(so)), 'evc_step_host')
        return out

    def gather_agent_obs(self, obs, delayed=None, out=None):
        """``[N, F]`` (+ optional delayed ``[N, F]``) -> materialised ``[N, n, F]`` per-agent
        observations on the device (multiagent_env.py:102-148)."""
        torch = self._torch()
        self._bind_stream()
        if out is None:
            out = torch.empty((self.N, self.n, self.F), dtype=torch.float32, device=obs.device)
        dptr = None if delayed is None else C.c_void_p(delayed.data_ptr())
        check(self.lib.evc_gather_agent_obs(self.handle, C.c_void_p(obs.data_ptr()), dptr,
                                            C.c_void_p(out.data_ptr())), 'evc_gather_agent_obs')
        return out

    def step_greedy(self, host: bool = True):
        """One step of the device-resident greedy policy (host buffers)."""
        out = self._host_buffers()
        so = self._step_out_struct(out, _np_ptr)
        check(self.lib.evc_step_host(self.handle, None, _lib.ACTION_GREEDY, 0, C.byref(so)), 'evc_step_host')
        return out

    def _pinned(self, shape, dtype) -> np.ndarray:
        """numpy-owned array, page-locked for the lifetime of the engine (evc_host_register): device<->host
        copies at PCIe speed.  Falls back to pageable memory if registration is refused."""
        arr = np.zeros(shape, dtype=dtype)
        if arr.nbytes and self.lib.evc_host_register(C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes)) == 0:
            self._registered.append(arr)
        return arr

    def _host_buffers(self) -> dict[str, np.ndarray]:
        """Two alternating sets of page-locked output arrays: what a step returns stays valid until the
        step after the next one (callers that keep observations longer copy them, as RL libraries do)."""
        if self._host_out is None:
            N, n, F = self.N, self.n, self.F
            sets = []
            for _ in range(2):
                out = {
                    'obs': self._pinned((N, F), np.float32), 'reward': self._pinned((N,), np.float64),
                    'terminated': self._pinned((N,), np.uint8), 'breakdown': self._pinned((N, 3), np.float64),
                    'final_obs': self._pinned((N, F), np.float32),
                }
                if self.debug_outputs:
                    for key in ('pilots', 'rates', 'projected'):
                        out[key] = self._pinned((N, n), np.float64)
                sets.append(out)
            self._host_out = sets
            self._host_flip = 0
        self._host_flip ^= 1
        return self._host_out[self._host_flip]

    def _step_host(self, actions: np.ndarray, bins: int):
        assert actions.shape == (self.N, self.n)
        if bins > 0:
            a = np.ascontiguousarray(actions, dtype=np.int64)
            kind = _lib.ACTION_DISCRETE
        else:
            a = np.ascontiguousarray(actions, dtype=np.float32)
            kind = _lib.ACTION_F32
        out = self._host_buffers()
        so = self._step_out_struct(out, _np_ptr)
        check(self.lib.evc_step_host(self.handle, _np_ptr(a), kind, bins, C.byref(so)), 'evc_step_host')
        return out

    def synchronize(self) -> None:
        check(self.lib.evc_synchronize(self.handle), 'evc_synchronize')

    def set_pipeline(self, halves: int) -> None:
        """``halves=2``: steps issued through ``make_stepper()`` run as two half-batch launches on two internal streams
        whose tails overlap the other half's next launch (``evc_set_pipeline``, include/evcharge.h).  Their outputs are
        complete on the torch stream only after ``join()``; ``step()`` / ``rollout()`` join by themselves."""
        check(self.lib.evc_set_pipeline(self.handle, int(halves)), 'evc_set_pipeline')
        self._pipeline = int(halves)

    def pipeline_halves(self):
        """Closed loop under ``set_pipeline(2)`` (``evc_pipeline_half``): ``[(slice, torch.cuda.ExternalStream), ...]`` for the
        two half batches.  A policy for half h — reading rows ``sl`` of the previous step's outputs, writing rows ``sl`` of
        the action tensor — enqueued under ``with torch.cuda.stream(stream):`` runs after that half's previous launch and
        before its next one, under the OTHER half's step; no ``join()`` between steps."""
        torch = self._torch()
        halves = []
        for h in range(2):
            st, lo, hi = C.c_void_p(), C.c_int32(), C.c_int32()
            check(self.lib.evc_pipeline_half(self.handle, h, C.byref(st), C.byref(lo), C.byref(hi)), 'evc_pipeline_half')
            halves.append((slice(lo.value, hi.value), torch.cuda.ExternalStream(st.value, device=torch.device('cuda', self.device))))
        return halves

    def join(self) -> None:
        """Orders the engine's (torch) stream after the pending half launches of the pipelined mode."""
        check(self.lib.evc_join(self.handle), 'evc_join')

    # ------------------------------------------------------------------ state access
    def env_scalars(self) -> dict[str, np.ndarray]:
        raw = np.zeros((self.N, 8), dtype=np.int32)
        check(self.lib.evc_get_env_scalars(self.handle, _np_ptr(raw)), 'evc_get_env_scalars')
        names = ('t', 'cursor', 'slot', 'moer_day', 'n_sessions', 'next_arrival', 'status', 'episodes')
        return {k: raw[:, i].copy() for i, k in enumerate(names)}

    def station_state(self):
        rem = np.zeros((self.N, self.n), np.float64)
        dep = np.zeros((self.N, self.n), np.int16)
        est = np.zeros((self.N, self.n), np.int16)
        check(self.lib.evc_get_station_state(self.handle, _np_ptr(rem), _np_ptr(dep), _np_ptr(est)),
              'evc_get_station_state')
        return rem, dep, est

    def get_state(self) -> dict[str, np.ndarray]:
        """Checkpoint of the whole simulator state (the reference has none)."""
        raw = np.zeros((self.N, 8), dtype=np.int32)
        check(self.lib.evc_get_env_scalars(self.handle, _np_ptr(raw)), 'evc_get_env_scalars')
        rem, dep, est = self.station_state()
        acc = np.zeros((self.N, 3), np.float64)
        check(self.lib.evc_get_breakdown(self.handle, _np_ptr(acc)), 'evc_get_breakdown')
        # order of the plugged-in EVs in each environment's entry list (compact layout): the delivered amps are summed in
        # that order, so it is part of a checkpoint that replays bit for bit (evcharge.h, ABI 7)
        rank = np.zeros((self.N, self.n), np.int16)
        check(self.lib.evc_get_entry_rank(self.handle, _np_ptr(rank)), 'evc_get_entry_rank')
        return {'scalars': raw, 'remaining_kwh': rem, 'departure': dep, 'est_departure': est,
                'breakdown': acc, 'entry_rank': rank}

    def set_state(self, state: dict[str, np.ndarray]) -> None:
        raw = np.ascontiguousarray(state['scalars'], dtype=np.int32)
        rem = np.ascontiguousarray(state['remaining_kwh'], dtype=np.float64)
        dep = np.ascontiguousarray(state['departure'], dtype=np.int16)
        est = np.ascontiguousarray(state['est_departure'], dtype=np.int16)
        acc = np.ascontiguousarray(state['breakdown'], dtype=np.float64)
        check(self.lib.evc_set_env_scalars(self.handle, _np_ptr(raw)), 'evc_set_env_scalars')
        rank = state.get('entry_rank')                # absent (older checkpoints, hand-made states): station order
        if rank is not None:
            rank = np.ascontiguousarray(rank, dtype=np.int16)
        check(self.lib.evc_set_station_state_ranked(self.handle, _np_ptr(rem), _np_ptr(dep), _np_ptr(est), _np_ptr(rank)),
              'evc_set_station_state_ranked')
        check(self.lib.evc_set_breakdown(self.handle, _np_ptr(acc)), 'evc_set_breakdown')

    def clear_status(self) -> None:
        check(self.lib.evc_clear_status(self.handle), 'evc_clear_status')

    def read_metrics(self) -> dict[str, float]:
        out = np.zeros(8, np.float64)
        check(self.lib.evc_read_metrics(self.handle, _np_ptr(out)), 'evc_read_metrics')
        return {'profit': out[0], 'carbon_cost': out[1], 'excess_charge': out[2],
                'env_steps': out[3], 'episodes_finished': out[4], 'envs_with_status': out[5],
                'solver_moved_values': out[6], 'tie_snap_near_boundary': out[7]}

    def set_tie_grid(self, log2_steps_per_amp: int = 16) -> None:
        """Grid of the tie snap, 2^-k A (``evc_set_tie_grid``; default 16).  k = 40 hands out the solvers'
        un-snapped optimum (the KKT certificate test)."""
        check(self.lib.evc_set_tie_grid(self.handle, int(log2_steps_per_amp)), 'evc_set_tie_grid')

    def last_slow_count(self) -> int:
        c = C.c_int32()
        check(self.lib.evc_last_slow_count(self.handle, C.byref(c)), 'evc_last_slow_count')
        return c.value

    def enable_timing(self, on: bool = True) -> None:
        check(self.lib.evc_enable_timing(self.handle, int(on)), 'evc_enable_timing')

    def last_step_ms(self) -> tuple[float, float]:
        a, b = C.c_float(), C.c_float()
        check(self.lib.evc_last_step_ms(self.handle, C.byref(a), C.byref(b)), 'evc_last_step_ms')
        return a.value, b.value

    def last_half_ms(self) -> tuple[float, float]:
        """Begin-to-end times of the two half launches of the last timed, pipelined step."""
        a, b = C.c_float(), C.c_float()
        check(self.lib.evc_last_half_ms(self.handle, C.byref(a), C.byref(b)), 'evc_last_half_ms')
        return a.value, b.value

    def pipelined_steps(self, ordered: bool = False):
        """Steps that ran as two half launches; with ``ordered=True`` also how many of them had to wait for work pending
        on the torch stream."""
        c, f = C.c_uint64(), C.c_uint64()
        check(self.lib.evc_pipelined_steps(self.handle, C.byref(c), C.byref(f)), 'evc_pipelined_steps')
        return (c.value, f.value) if ordered else c.value
