"""Batched battery-dispatch environment (BASELINE config 4, "ElectricityMarketEnv battery-dispatch
step") on the HIP engine of include/battery_dispatch.h.

A SYNTHETIC WORKLOAD: the reference snapshot has no ElectricityMarketEnv code
(docs/electricitymarketenv.md:3-27 is the only specification; sustaingym/envs/battery_storage.py is a
NotImplementedError stub), so there is nothing to be identical to.  Observation / action / reward
follow the prose; the market clearing (a multi-period SCED LP upstream) is replaced by a price-taker
rule on supplied price traces.  See DESIGN.md §10.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


EPISODE_STEPS, TRACE_LEN = 288, 289


def synthetic_market_traces(num_episodes: int, k: int = 36, seed: int = 0):
    """Price / load / MOER day traces of the shape the prose describes (5-minute settlements, a
    duck-curve-like price, forecasts = the series itself plus noise).  Returns a dict of float32
    arrays: price, load, moer [P, 289]; load_fc, moer_fc [P, 289 + k]; terminal_price [P] float64."""
    rng = np.random.default_rng(seed)
    P = num_episodes
    t = np.arange(TRACE_LEN + k)[None, :] / 288.0
    phase = rng.uniform(-0.05, 0.05, (P, 1))
    load = 3000 + 900 * np.sin(2 * np.pi * (t - 0.3 + phase)) + 350 * np.sin(4 * np.pi * (t + phase)) + 40 * rng.standard_normal((P, TRACE_LEN + k))
    price = 35 + 25 * np.sin(2 * np.pi * (t - 0.32 + phase)) + 18 * np.maximum(0, np.sin(4 * np.pi * (t - 0.1))) + 3 * rng.standard_normal((P, TRACE_LEN + k))
    price = np.maximum(price, 1.0)
    moer = 380 + 120 * np.sin(2 * np.pi * (t - 0.45 + phase)) + 10 * rng.standard_normal((P, TRACE_LEN + k))
    out = {
        'price': price[:, :TRACE_LEN], 'load': load[:, :TRACE_LEN], 'moer': moer[:, :TRACE_LEN],
        'load_fc': load + 25 * rng.standard_normal(load.shape), 'moer_fc': moer + 8 * rng.standard_normal(moer.shape),
    }
    out = {key: np.ascontiguousarray(v, dtype=np.float32) for key, v in out.items()}
    out['terminal_price'] = out['price'].astype(np.float64).mean(axis=1)
    return out


class BatteryDispatchVectorEnv:
    """N independent battery-dispatch environments stepped by one kernel launch.

    ``reset(slots=None) -> obs[N, 4k+6]``; ``step(bids[N, 2k]) -> (obs, reward[N], terminated[N])``;
    numpy in / numpy out (host staging; fresh arrays every call) or CUDA tensors in / out (``output='torch'``).

    With ``output='torch'`` the three tensors returned by ``reset`` / ``step`` are the environment's PERSISTENT device
    buffers — the same ``obs``, ``reward`` and ``terminated`` (bool) objects every call, overwritten by the next step on
    the stream the call was issued on: a caller that keeps a step's values clones them."""

    def __init__(self, num_envs: int, forecast_steps: int = 36, bank_slots: int | None = None, device: int = 0,
                 capacity_mwh: float = 80.0, max_power_mw: float = 20.0, eta_charge: float = 0.95,
                 eta_discharge: float = 0.95, init_energy_mwh: float = 40.0, co2_price_per_kg: float = 0.03085,
                 output: str = 'numpy'):
        assert output in ('numpy', 'torch')
        self.lib = _lib.load()
        self.N, self.k = int(num_envs), int(forecast_steps)
        self.F = 4 * self.k + 6
        self.bank_slots = int(bank_slots or num_envs)
        self.device, self.output = device, output
        cfg = _lib.BatConfig(self.N, self.k, self.bank_slots, device, capacity_mwh, max_power_mw, eta_charge,
                             eta_discharge, init_energy_mwh, co2_price_per_kg)
        h = C.c_void_p()
        self._check(self.lib.bat_create(C.byref(cfg), C.byref(h)), 'bat_create')
        self.handle = h
        self._obs = np.zeros((self.N, self.F), np.float32)
        self._rew = np.zeros(self.N, np.float64)
        self._term = np.zeros(self.N, np.uint8)
        self._dev = None
        self._owned_traj: dict = {}       # data_ptr of a trajectory view made by new_trajectory() -> its padded allocation

    def _check(self, rc, what):
        if rc != 0:
            raise _lib.EngineLibraryError(f'{what} failed with code {rc}: {self.lib.bat_last_error().decode()}')

    def upload_traces(self, traces: dict, first_slot: int = 0) -> None:
        arrs = [np.ascontiguousarray(traces[key], dtype=np.float32) for key in ('price', 'load', 'load_fc', 'moer', 'moer_fc')]
        tp = np.ascontiguousarray(traces['terminal_price'], dtype=np.float64)
        count = arrs[0].shape[0]
        assert arrs[0].shape == (count, TRACE_LEN) and arrs[2].shape == (count, TRACE_LEN + self.k)
        self._check(self.lib.bat_upload_traces(self.handle, first_slot, count, arrs[0].ctypes.data, arrs[1].ctypes.data,
                                               arrs[2].ctypes.data, arrs[3].ctypes.data, arrs[4].ctypes.data,
                                               tp.ctypes.data), 'bat_upload_traces')

    def _device_buffers(self):
        if self._dev is None:
            import torch
            dev = torch.device('cuda', self.device)
            self._dev = (torch.zeros((self.N, self.F), dtype=torch.float32, device=dev),
                         torch.zeros(self.N, dtype=torch.float64, device=dev),
                         torch.zeros(self.N, dtype=torch.bool, device=dev))      # the kernel stores 0 / 1 bytes: a bool tensor as it is
        import torch
        self._check(self.lib.bat_set_stream(self.handle, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                    'bat_set_stream')
        return self._dev

    def reset(self, slots=None):
        sl = None if slots is None else np.ascontiguousarray(slots, dtype=np.int32)
        sp = None if sl is None else sl.ctypes.data
        if self.output == 'torch':
            obs, _, _ = self._device_buffers()
            self._check(self.lib.bat_reset(self.handle, sp, C.c_void_p(obs.data_ptr())), 'bat_reset')
            return obs
        self._check(self.lib.bat_reset_host(self.handle, sp, self._obs.ctypes.data), 'bat_reset_host')
        return self._obs.copy()

    def step(self, bids):
        if self.output == 'torch':
            obs, rew, term = self._device_buffers()
            assert bids.is_cuda and bids.dtype.is_floating_point and tuple(bids.shape) == (self.N, 2 * self.k)
            b = bids.contiguous().float()
            self._check(self.lib.bat_step(self.handle, C.c_void_p(b.data_ptr()), C.c_void_p(obs.data_ptr()),
                                          C.c_void_p(rew.data_ptr()), C.c_void_p(term.data_ptr())), 'bat_step')
            return obs, rew, term            # (a `.bool()` here was a second kernel launch per step: 3 of round 2's 10.6 us)
        b = np.ascontiguousarray(bids, dtype=np.float32)
        assert b.shape == (self.N, 2 * self.k)
        self._check(self.lib.bat_step_host(self.handle, b.ctypes.data, self._obs.ctypes.data, self._rew.ctypes.data,
                                           self._term.ctypes.data), 'bat_step_host')
        return self._obs.copy(), self._rew.copy(), self._term.astype(bool)

    def new_trajectory(self, steps: int):
        """``(obs_traj [steps, N, 4k+6] float32, reward_traj [steps, N] float64)`` for ``rollout(..., out=...)``, allocated the way
        ``rollout`` does it itself: observation rows 640 B apart (a whole number of 128-byte lines), zero-filled, the ``[:, :, :4k+6]``
        view handed out.  The padding behind each row belongs to the kernel (whole-line non-temporal stores): buffers made here
        are remembered, any OTHER strided view passed as ``out`` only ever has its 4k+6 floats per row written."""
        import torch
        obs, _, _ = self._device_buffers()
        pitch = (self.F + 31) // 32 * 32
        store = torch.zeros((steps, self.N, pitch), dtype=torch.float32, device=obs.device)
        view = store[:, :, :self.F]
        self._owned_traj[view.data_ptr()] = store            # (keeps the allocation alive as long as the environment)
        return view, torch.zeros((steps, self.N), dtype=torch.float64, device=obs.device)

    def rollout(self, bids_ring, steps: int, trajectory: bool = False, out=None):
        """``steps`` steps in ONE launch (``bat_rollout``): step i uses ``bids_ring[i % R]`` (CUDA float32 ``[R, N, 2k]``).
        Returns ``(obs, reward, terminated)`` of the last step (the persistent buffers of ``step``) and, with
        ``trajectory=True``, also ``(obs_traj [steps, N, 4k+6] float32, reward_traj [steps, N] float64)`` — every step's
        observation and reward, what a learner's rollout buffer holds (pass ``out=(obs_traj, reward_traj)`` to reuse buffers)."""
        import torch
        assert self.output == 'torch' and bids_ring.is_cuda and bids_ring.dtype == torch.float32 and bids_ring.is_contiguous()
        assert bids_ring.dim() == 3 and tuple(bids_ring.shape[1:]) == (self.N, 2 * self.k)
        obs, rew, term = self._device_buffers()
        traj = None
        pitch = self.F
        if trajectory:
            if out is None:
                # rows 640 B apart (a multiple of the 128-byte line; bat_rollout_pitched): the [steps, N, 4k+6] VIEW is handed out.
                # Zero-filled: the rows of steps after an environment's termination are not written by the kernel.
                traj = self.new_trajectory(steps)
                del self._owned_traj[traj[0].data_ptr()]      # handed out for good: nothing to remember
                pitch = -int(traj[0].stride(1))   # the padding of OUR allocation is the kernel's to fill: whole-line stores (battery_dispatch.h)
            else:
                traj = out
                assert tuple(traj[0].shape) == (steps, self.N, self.F) and tuple(traj[1].shape) == (steps, self.N)
                assert traj[0].stride(2) == 1 and traj[0].stride(0) == self.N * traj[0].stride(1), 'obs_traj: rows at a constant pitch'
                pitch = int(traj[0].stride(1))   # positive: nothing behind the 4k+6 floats of a row is written (a caller's wider tensor)
                if traj[0].data_ptr() in self._owned_traj and traj[0].shape[0] <= self._owned_traj[traj[0].data_ptr()].shape[0]:
                    pitch = -pitch               # a buffer new_trajectory() made: its padding is the kernel's
        self._check(self.lib.bat_rollout_pitched(self.handle, C.c_void_p(bids_ring.data_ptr()), int(bids_ring.shape[0]), int(steps),
                                                 C.c_void_p(obs.data_ptr()), C.c_void_p(rew.data_ptr()), C.c_void_p(term.data_ptr()),
                                                 C.c_void_p(traj[0].data_ptr()) if traj else None, pitch,
                                                 C.c_void_p(traj[1].data_ptr()) if traj else None), 'bat_rollout')
        return (obs, rew, term) + ((traj,) if traj else ())

    def make_stepper(self):
        """Lean per-step callable for throughput loops (the counterpart of ``StepEngine.make_stepper``): binds the current
        torch stream and the output buffers once and returns ``(step(ptr: int) -> None, (obs, reward, terminated))`` where
        ``ptr`` is the device address of a contiguous float32 ``[N, 2k]`` bid tensor.  ``step()`` above spends more host
        time on checks and conversions (12.7 us) than the kernel runs (7.4 us)."""
        assert self.output == 'torch'
        obs, rew, term = self._device_buffers()
        fn, handle = self.lib.bat_step, self.handle
        po, pr, pt = C.c_void_p(obs.data_ptr()), C.c_void_p(rew.data_ptr()), C.c_void_p(term.data_ptr())

        def step(ptr: int) -> None:
            rc = fn(handle, ptr, po, pr, pt)
            if rc:
                self._check(rc, 'bat_step')
        return step, (obs, rew, term)

    def state(self):
        e = np.zeros(self.N, np.float64)
        t = np.zeros(self.N, np.int32)
        self._check(self.lib.bat_get_state(self.handle, e.ctypes.data, t.ctypes.data), 'bat_get_state')
        return e, t

    def read_metrics(self) -> dict:
        out = np.zeros(4, np.float64)
        self._check(self.lib.bat_read_metrics(self.handle, out.ctypes.data), 'bat_read_metrics')
        return {'energy_mwh': out[0], 'returns': out[1], 'env_steps': out[2], 'terminated': out[3]}

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.bat_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
