"""Drop-in environment classes over the HIP step engine.

Same names, constructor arguments, spaces, return conventions and error behaviour as the
reference's API layer:

* :class:`EVChargingEnv`            <- sustaingym/envs/evcharging/env.py:20 (gymnasium.Env)
* :class:`DiscreteActionWrapper`    <- sustaingym/envs/wrappers.py:13
* :class:`MultiAgentEVChargingEnv`  <- sustaingym/envs/evcharging/multiagent_env.py:18 (PettingZoo Parallel)
* :class:`EVChargingVectorEnv`      <- what SB3 ``SubprocVecEnv`` / RLLib rollout workers provide
  around the reference env (train_stable_baselines.py:271-275): N environments, Gymnasium-0.28
  ``VectorEnv`` semantics (autoreset, ``final_observation``), backed by ONE engine call per step.
* :class:`SB3VecEnv`                <- stable_baselines3 ``VecEnv`` protocol adapter.
* :class:`RLlibVectorEnv`           <- ``ray.rllib.env.VectorEnv`` protocol adapter (what ``num_envs_per_worker`` builds out of
  copies of the reference env, examples/evcharging/train_rllib.py:129-134,158-160).

All arithmetic of ``step()`` runs in the HIP kernels through the C-ABI; these classes only move
buffers and keep host-side episode bookkeeping (generators, max_profit).
"""
from __future__ import annotations

from typing import Any, Callable, Sequence

import numpy as np

from . import spaces
from .engine import StepEngine
from .event_generation import (AbstractTraceGenerator, BatchedGMMTraceGenerator, DeviceGMMTraceGenerator,
                               EventTable, RealTraceBank)
from .network import site_str_to_site

# With gymnasium / pettingzoo installed the classes derive from the same bases as the reference's
# (env.py:20 gymnasium.Env, wrappers.py:13 gymnasium.ActionWrapper, multiagent_env.py:18 ParallelEnv), so
# isinstance checks of StableBaselines3 / RLLib's PettingZoo adapter hold; without them (this build
# image) they are plain classes with the same protocol.
try:  # pragma: no cover
    import gymnasium as _gym
    _EnvBase = _gym.Env
    _ActionWrapperBase = _gym.ActionWrapper
except Exception:
    _gym = None
    _EnvBase = object
    _ActionWrapperBase = object
try:  # pragma: no cover
    from gymnasium.vector import VectorEnv as _VectorEnvBase
    from gymnasium.vector.utils import batch_space as _batch_space
except Exception:
    _VectorEnvBase = object
    _batch_space = None
try:  # pragma: no cover
    from pettingzoo import ParallelEnv as _ParallelEnvBase
except Exception:
    _ParallelEnvBase = object
try:  # pragma: no cover
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnvBase
except Exception:
    _SB3VecEnvBase = object
try:  # pragma: no cover
    from ray.rllib.env.vector_env import VectorEnv as _RLlibVectorEnvBase
except Exception:
    _RLlibVectorEnvBase = object

MAX_SESSIONS = 256
OBS_KEYS = ('demands', 'est_departures', 'forecasted_moer', 'prev_moer', 'timestep')  # sorted


def obs_slices(n: int, k: int) -> dict[str, slice]:
    """Position of each Dict key inside the flattened observation row (gymnasium sorts keys)."""
    return {'demands': slice(0, n), 'est_departures': slice(n, 2 * n),
            'forecasted_moer': slice(2 * n, 2 * n + k), 'prev_moer': slice(2 * n + k, 2 * n + k + 1),
            'timestep': slice(2 * n + k + 1, 2 * n + k + 2)}


def make_observation_space(n: int, k: int, requested_energy_cap: float) -> spaces.Dict:
    """env.py:143-150."""
    return spaces.Dict({
        'timestep': spaces.Box(0, 1, shape=(1,), dtype=np.float32),
        'est_departures': spaces.Box(-288, 288, shape=(n,), dtype=np.float32),
        'demands': spaces.Box(0, requested_energy_cap, shape=(n,), dtype=np.float32),
        'prev_moer': spaces.Box(0, 1, shape=(1,), dtype=np.float32),
        'forecasted_moer': spaces.Box(0, 1, shape=(k,), dtype=np.float32),
    })


def _pad_table(table: EventTable, stride: int):
    from ._lib import SESSION_DTYPE
    if len(table) > stride:
        raise ValueError(f'episode has {len(table)} sessions, engine capacity is {stride}')
    s = np.zeros(stride, dtype=SESSION_DTYPE)
    r = np.zeros(stride, dtype=np.float64)
    s[:len(table)] = table.sessions
    r[:len(table)] = table.requested
    return s, r


class EVChargingEnv(_EnvBase):
    """Single-agent EV charging environment (reference: env.py:20-470), one environment on the
    GPU engine.  See the reference class docstring for the MDP; constants below are env.py:99-114."""
    TIMESTEP_DURATION = 5
    ACTION_SCALE_FACTOR = 32
    VOLTAGE = 208
    MARGINAL_REVENUE_PER_KWH = 0.15
    OPERATING_MARGIN = 0.20
    MARGINAL_PROFIT_PER_KWH = MARGINAL_REVENUE_PER_KWH * OPERATING_MARGIN
    CO2_COST_PER_METRIC_TON = 30.85
    A_MINS_TO_KWH = (1 / 60) * (VOLTAGE / 1000)
    VIOLATION_WEIGHT = 0.001
    A_PERS_TO_KWH = A_MINS_TO_KWH * TIMESTEP_DURATION
    PROFIT_FACTOR = A_PERS_TO_KWH * MARGINAL_PROFIT_PER_KWH
    VIOLATION_FACTOR = A_PERS_TO_KWH * VIOLATION_WEIGHT
    CARBON_COST_FACTOR = A_PERS_TO_KWH * (CO2_COST_PER_METRIC_TON / 1000)

    metadata: dict[str, Any] = {}
    render_mode = None

    def __init__(self, data_generator: AbstractTraceGenerator, moer_forecast_steps: int = 36,
                 project_action_in_env: bool = True, verbose: int = 0, device: int = 0,
                 charge_calculation: str = 'continuous'):
        assert 1 <= moer_forecast_steps <= 36                                   # env.py:120
        self.data_generator = data_generator
        self.max_timestep = 288                                                  # env.py:124
        self.moer_forecast_steps = moer_forecast_steps
        self.project_action_in_env = project_action_in_env
        self.verbose = verbose
        self.cn = site_str_to_site(data_generator.site)                          # env.py:132
        self.num_stations = len(self.cn.station_ids)
        n, k = self.num_stations, moer_forecast_steps
        self.observation_space = make_observation_space(n, k, data_generator.requested_energy_cap)
        self.action_space = spaces.Box(low=0, high=1.0, shape=(n,), dtype=np.float32)  # env.py:171
        self._engine = StepEngine(self.cn, 1, moer_forecast_steps=k, project_action=project_action_in_env,
                                  autoreset=False, device=device, bank_slots=1,
                                  max_sessions=MAX_SESSIONS, moer_days=1, debug_outputs=True,
                                  charge_calculation=charge_calculation)
        self._flat = np.zeros(2 * n + k + 2, dtype=np.float32)
        sl = obs_slices(n, k)
        # like the reference (env.py:152-158) the observation arrays are reused buffers
        self._obs = {key: self._flat[sl[key]] for key in
                     ('timestep', 'est_departures', 'demands', 'prev_moer', 'forecasted_moer')}
        self._reward_breakdown = {'profit': 0.0, 'carbon_cost': 0.0, 'excess_charge': 0.0}
        self._max_profit = 0.0
        self._evs: EventTable | None = None
        self._pilots = np.zeros((289, n))
        self.moer: np.ndarray | None = None
        self.t = 0
        self._needs_reset = True

    def __repr__(self) -> str:
        return (f'EVChargingGym (action projection = {self.project_action_in_env}, '
                f'moer forecast steps = {self.moer_forecast_steps}) '
                f'using {self.data_generator.__repr__()}')

    # -- reference API --------------------------------------------------------------------
    def reset(self, *, seed: int | None = None, options: dict[str, Any] | None = None):
        """env.py:293-338."""
        if _gym is not None:  # pragma: no cover
            super().reset(seed=seed)
        self.data_generator.set_seed(seed)                                       # env.py:314
        if options is not None and 'verbose' in options:
            self.verbose = options['verbose']
        table = self.data_generator.get_event_table()                            # env.py:321
        self._evs = table
        self._max_profit = table.max_profit()                                    # env.py:322
        self.moer = self.data_generator.get_moer()                               # env.py:323 (advanced day)
        s, r = _pad_table(table, MAX_SESSIONS)
        self._engine.upload_moer(self.moer[None], 0)
        self._engine.upload_episodes([len(table)], s[None], r[None], [0], 0)
        obs = self._engine.reset(host=True)
        self._flat[:] = obs[0]
        self.t = 0
        for key in self._reward_breakdown:
            self._reward_breakdown[key] = 0.0
        self._pilots[:] = 0
        self._needs_reset = False
        if self.verbose >= 1:
            print(f'Simulating {len(table)} events using {self.data_generator}')
        return self._obs, self._get_info()

    def step(self, action: np.ndarray):
        """env.py:229-291."""
        if self._needs_reset:
            raise RuntimeError('call reset() before step()')      # reference: AttributeError on _simulator
        action = np.asarray(action)
        if np.issubdtype(action.dtype, np.integer):
            raise TypeError('integer actions need DiscreteActionWrapper')
        out = self._engine.step(np.ascontiguousarray(action, dtype=np.float32).reshape(1, -1))
        self.t += 1
        self._flat[:] = out['obs'][0]
        reward = float(out['reward'][0])
        done = bool(out['terminated'][0])
        bd = out['breakdown'][0]
        self._reward_breakdown['profit'] = float(bd[0])
        self._reward_breakdown['carbon_cost'] = float(bd[1])
        self._reward_breakdown['excess_charge'] = float(bd[2])
        self._pilots[self.t - 1] = out['pilots'][0]
        self._last = out
        if done:
            self._needs_reset = True
        return self._obs, reward, done, False, self._get_info()

    def _get_info(self, all: bool = False) -> dict[str, Any]:                   # env.py:396-416
        info = {'max_profit': self._max_profit, 'reward_breakdown': self._reward_breakdown}
        if all:
            info.update({'num_evs': len(self._evs), 'avg_plugin_time': self._evs.avg_plugin_time(),
                         'evs': self._evs, 'moer': self.moer,
                         'pilot_signals': self._pilots[:self.t].copy(),
                         'status': int(self._engine.env_scalars()['status'][0])})
        return info

    def close(self) -> None:                                                     # env.py:466-470
        if getattr(self, '_engine', None) is not None:
            self._engine.close()
            self._engine = None

    def render(self) -> None:
        return None


class DiscreteActionWrapper(_ActionWrapperBase):
    """wrappers.py:13-45: discrete {0..bins-1}^n actions -> a/(bins-1).  The float32 division
    itself runs in the engine (EVC_ACTION_DISCRETE)."""

    def __init__(self, env: EVChargingEnv, bins: int = 5):
        if not isinstance(env.action_space, spaces.Box):
            raise ValueError('Should only be used to wrap continuous env')      # wrappers.py:28
        if _ActionWrapperBase is not object:
            super().__init__(env)
        self.env = env
        self._bins = bins
        dims = env.action_space.shape
        self.action_space = (spaces.Discrete(bins) if len(dims) == 0 else
                             spaces.MultiDiscrete(np.ones(dims, dtype=np.int64) * bins))
        self.observation_space = env.observation_space

    def __repr__(self) -> str:
        return repr(self.env)

    def __getattr__(self, name):
        if name == 'env':                          # not constructed yet: no recursion
            raise AttributeError(name)
        return getattr(self.env, name)

    def action(self, action):                                                    # wrappers.py:43-45
        return np.asarray(action, dtype=np.float32) / (self._bins - 1)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, action):
        return self.env.step(self.action(action))

    def close(self):
        return self.env.close()


class MultiAgentEVChargingEnv(_ParallelEnvBase):
    """multiagent_env.py:18-218 (PettingZoo ``ParallelEnv`` protocol, one agent per EVSE).

    ``periods_delay`` > 0: the reference implementation stores aliases of reused buffers, so its
    delayed observations equal the current ones.  The default
    ``delay_semantics='reference'`` reproduces that; ``'documented'`` implements the documented
    behaviour (own information current, other agents' ``periods_delay`` steps old)."""
    metadata: dict[str, Any] = {}

    def __init__(self, data_generator: AbstractTraceGenerator, periods_delay: int = 0,
                 moer_forecast_steps: int = 36, project_action_in_env: bool = True,
                 discrete_bins: int = -1, verbose: int = 0, delay_semantics: str = 'reference',
                 device: int = 0):
        assert delay_semantics in ('reference', 'documented')
        self.periods_delay = periods_delay
        self.delay_semantics = delay_semantics
        base = EVChargingEnv(data_generator, moer_forecast_steps, project_action_in_env, verbose, device)
        self._base = base
        self.single_env = base if discrete_bins <= 0 else DiscreteActionWrapper(base, bins=discrete_bins)
        self.agents = base.cn.station_ids[:]
        self.possible_agents = self.agents
        flat_space = spaces.flatten_space(base.observation_space)                # :88
        self.observation_spaces = {a: flat_space for a in self.agents}
        act = spaces.Discrete(discrete_bins) if discrete_bins > 0 else spaces.Box(0., 1., shape=(1,))
        self.action_spaces = {a: act for a in self.agents}
        self._discrete = discrete_bins > 0
        self._past: list[np.ndarray] = []

    @property
    def num_agents(self) -> int:
        return len(self.agents)

    @property
    def max_num_agents(self) -> int:
        return len(self.possible_agents)

    def _obs_dict(self, init: bool = False) -> dict[str, np.ndarray]:
        flat = self._base._flat.copy()                  # spaces.flatten(...) of the Dict obs (:115)
        if self.periods_delay == 0 or self.delay_semantics == 'reference':
            return {a: flat for a in self.agents}
        n = self._base.num_stations
        if init:
            self._past = [flat.copy() for _ in range(self.periods_delay)]
            return {a: flat for a in self.agents}
        old = self._past.pop(0)
        self._past.append(flat.copy())
        out = {}
        for i, a in enumerate(self.agents):
            o = flat.copy()
            o[:2 * n] = old[:2 * n]                     # others: delayed demands / est_departures
            o[i], o[n + i] = flat[i], flat[n + i]       # own: current
            out[a] = o
        return out

    def step(self, actions: dict[str, np.ndarray]):
        """multiagent_env.py:157-195."""
        dtype = np.int64 if self._discrete else np.float32
        action = np.zeros(len(self.agents), dtype=dtype)
        for i, a in enumerate(self.agents):
            action[i] = np.asarray(actions[a]).reshape(-1)[0] if not np.isscalar(actions[a]) else actions[a]
        obs, reward, terminated, truncated, info = self.single_env.step(action)
        obss = self._obs_dict()
        n_agents = self.num_agents
        rewards = {a: float(reward) / n_agents for a in self.agents}            # :186
        terminateds = {a: terminated for a in self.agents}
        truncateds = {a: truncated for a in self.agents}
        infos = {a: info for a in self.agents}
        if terminated or truncated:
            self.agents = []                                                     # :192-193
        return obss, rewards, terminateds, truncateds, infos

    def reset(self, seed: int | None = None, options: dict | None = None):
        _, info = self.single_env.reset(seed=seed, options=options)
        self.agents = self.possible_agents[:]
        return self._obs_dict(init=True), {a: info for a in self.agents}

    def render(self) -> None:
        return None

    def close(self) -> None:
        self.single_env.close()

    def observation_space(self, agent: str):
        return self.observation_spaces[agent]

    def action_space(self, agent: str):
        return self.action_spaces[agent]


class EVChargingVectorEnv(_VectorEnvBase):
    """N independent EVChargingEnv instances stepped by one engine call (Gymnasium 0.28
    ``VectorEnv`` semantics: batched dict observation, autoreset, ``final_observation``).

    ``data_generators``: a list of N generators or a factory ``i -> generator`` (like the
    ``env_fns`` of SB3's SubprocVecEnv, train_stable_baselines.py:271-275).  All generators must
    cover the same site and date period (the period's MOER days are uploaded once).  Episodes are
    double-buffered in the engine's bank: while environment i plays the episode in slot i the next
    one already sits in slot i+N, and the kernel-side autoreset flips between the two.

    For large N pass one :class:`BatchedGMMTraceGenerator` (plus ``num_envs``): all episodes of a
    boundary are drawn in one vectorised call on a worker thread while the GPU plays the current
    episodes (~50 us per episode instead of ~2 ms through N per-environment generators).  With a
    :class:`DeviceGMMTraceGenerator` the bank is refilled by the GPU itself (``evc_generate_episodes``,
    ~10 ns per episode) and only the ``max_profit`` values travel to the host.  A :class:`RealTraceBank`
    keeps every real day of the period resident and walks the days sequentially inside the kernel.

    ``output='numpy'`` (default) returns host arrays (SB3 / RLLib) — copies, unless ``zero_copy=True``
    hands out the engine's two alternating page-locked buffer sets (valid until the step after the next
    one; what SB3VecEnv, which copies anyway, and throughput measurements use).  ``'torch'`` takes and
    returns device tensors without leaving the GPU; those ARE the engine's output buffers and are
    overwritten by the next step (clone what must be kept).  On that path the observation dict, the
    ``reward_breakdown`` dict and — within an episode — the ``info`` dict are the SAME objects every step
    (like the reference's reused observation buffers, env.py:152-158): a caller that stores or mutates them
    sees them change; copy what must be kept.

    ``pipeline=2`` (``output='torch'`` only; opt-in, changes the ordering contract): float32 steps run as two
    half-batch launches on two internal streams (``evc_set_pipeline``) and ``step()`` does NOT order the caller's
    stream after them.  A closed-loop policy keeps the overlap by computing each half's actions on that half's stream::

        halves = venv.pipeline_halves()                  # [(slice, torch.cuda.ExternalStream)] x 2
        for sl, stream in halves:
            with torch.cuda.stream(stream):
                actions[sl] = policy(obs['demands'][sl])  # rows of this half only
        obs, rew, term, trunc, info = venv.step(actions)

    so that the policy of one half runs under the other half's step; ``venv.join()`` orders the current stream after
    both halves (before reading a whole-batch output there)."""

    def __init__(self, data_generators: Sequence[AbstractTraceGenerator] | Callable[[int], AbstractTraceGenerator],
                 num_envs: int | None = None, moer_forecast_steps: int = 36,
                 project_action_in_env: bool = True, discrete_bins: int = -1, device: int = 0,
                 output: str = 'numpy', max_sessions: int = 128, charge_calculation: str = 'continuous',
                 zero_copy: bool = False, pipeline: int = 1):
        assert output in ('numpy', 'torch')
        assert pipeline in (1, 2) and (pipeline == 1 or output == 'torch'), 'pipeline=2 needs output="torch"'
        self.pipeline = int(pipeline)
        self.zero_copy = bool(zero_copy)
        self._batched = data_generators if isinstance(data_generators, BatchedGMMTraceGenerator) else None
        self._devgen = data_generators if isinstance(data_generators, DeviceGMMTraceGenerator) else None
        self._realbank = data_generators if isinstance(data_generators, RealTraceBank) else None
        if self._batched is not None or self._devgen is not None or self._realbank is not None:
            assert num_envs is not None, 'num_envs is required with a batched / device generator'
            gens = [data_generators] * num_envs
        elif callable(data_generators):
            assert num_envs is not None
            gens = [data_generators(i) for i in range(num_envs)]
        else:
            gens = list(data_generators)
        self.generators = gens
        self.num_envs = N = len(gens)
        g0 = gens[0]
        assert all(g.site == g0.site and g.date_range_str == g0.date_range_str for g in gens), \
            'all generators must share site and date period'
        self.cn = site_str_to_site(g0.site)
        self.num_stations = n = self.cn.num_stations
        self.moer_forecast_steps = k = moer_forecast_steps
        self.project_action_in_env = project_action_in_env
        self.discrete_bins = discrete_bins
        self.output = output
        self._stride = max_sessions
        self.single_observation_space = make_observation_space(n, k, g0.requested_energy_cap)
        self.single_action_space = (spaces.MultiDiscrete(np.full(n, discrete_bins, np.int64))
                                    if discrete_bins > 0 else
                                    spaces.Box(low=0, high=1.0, shape=(n,), dtype=np.float32))
        # gymnasium.vector.VectorEnv's attributes (its __init__ is not called: it would only set these)
        self.observation_space = (_batch_space(self.single_observation_space, N) if _batch_space
                                  else self.single_observation_space)
        self.action_space = (_batch_space(self.single_action_space, N) if _batch_space
                             else self.single_action_space)
        self.is_vector_env = True
        self._ndays = g0.num_days_in_date_range
        self._day0 = g0.date_range[0]
        if self._realbank is not None:
            max_sessions = self._stride = self._realbank.sessions.shape[1]
        self._bank_slots = self._ndays if self._realbank is not None else 2 * N
        self._engine = StepEngine(self.cn, N, moer_forecast_steps=k, project_action=project_action_in_env,
                                  autoreset=True, device=device, bank_slots=self._bank_slots,
                                  max_sessions=max_sessions, moer_days=self._ndays,
                                  charge_calculation=charge_calculation)
        from datetime import timedelta
        moer = np.stack([g0.moer_loader.retrieve(self._day0 + timedelta(days=d)) for d in range(self._ndays)])
        self._engine.upload_moer(moer, 0)
        self._engine.set_autoreset_stride(1 if self._realbank is not None else N)
        if self.pipeline == 2:
            self._engine.set_pipeline(2)
        if self._devgen is not None:
            self._engine.upload_gmm(self._devgen.tables)
        if self._realbank is not None:                 # the whole period, once
            b = self._realbank
            self._engine.upload_episodes(b.n_sessions, b.sessions, b.requested, b.moer_day, 0)
        self._slices = obs_slices(n, k)
        self._max_profit = (self._realbank.max_profit.copy() if self._realbank is not None
                            else np.zeros(2 * N))     # per bank slot
        self._cur_slot = np.arange(N)                 # slot each env is playing
        self._episodes = np.zeros(N, dtype=np.int64)
        self._steps_in_episode = 0                    # all environments run in lock-step (288 steps)
        self._pending = None                          # (future, slots): background refill, batched mode
        self._max_profit_pending: list[tuple[int, int]] = []   # device-generated slots whose max_profit is still on the GPU
        self._info_max_profit = None                  # cached per episode
        self._false_dev = None
        self._stepper = None
        self._stepper_out = None
        self._stepper_stream = None
        self._stepper_policy = None
        self._lean_views = None
        self._lean_info = None
        self._pool = None
        self.closed = False

    # -- episode staging ------------------------------------------------------------------
    def _upload_runs(self, slots, ns, sess, req, day) -> None:
        order = np.argsort(slots)
        s_sorted = slots[order]
        start = 0
        for end in range(1, len(slots) + 1):           # contiguous runs of slots upload in one call each
            if end == len(slots) or s_sorted[end] != s_sorted[end - 1] + 1:
                sel = order[start:end]
                self._engine.upload_episodes(ns[sel], sess[sel], req[sel], day[sel], int(s_sorted[start]))
                start = end

    def _stage_batched(self, slots: np.ndarray, background: bool = False) -> None:
        """One vectorised draw for all ``slots``; with ``background`` the sampling runs on a worker
        thread while the GPU keeps stepping and is uploaded by :meth:`_drain` (at the latest before
        the step that ends the current episodes)."""
        def work():
            return self._batched.sample_episodes(len(slots), self._stride)
        if background:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=1)
            self._pending = (self._pool.submit(work), slots)
        else:
            self._finish_batched(slots, work())

    def _stage_device(self, slots: np.ndarray) -> None:
        """Fills ``slots`` on the GPU (one launch per contiguous run); only the max_profit values of
        the new episodes come back to the host."""
        g = self._devgen
        s_sorted = np.sort(slots)
        # contiguous runs, found in numpy (a Python loop over 65 536 slots cost 9 ms per episode boundary: 30 us per step of the
        # API's whole-episode average, profiles/r5_venv_blocks.txt)
        cuts = np.flatnonzero(np.diff(s_sorted) != 1) + 1
        starts = np.concatenate(([0], cuts))
        ends = np.concatenate((cuts, [len(s_sorted)]))
        self._flush_max_profit()
        for start, end in zip(starts.tolist(), ends.tolist()):
            first, cnt = int(s_sorted[start]), end - start
            self._engine.generate_episodes(first, cnt, g.seed, g.next_episode)
            g.next_episode += cnt
            # max_profit of these episodes is first needed when they are PLAYED (one episode from now, or at once after
            # reset()): fetched then (_flush_max_profit), not behind the generating kernel with the GPU idle meanwhile
            self._max_profit_pending.append((first, cnt))

    def _flush_max_profit(self, needed: np.ndarray | None = None) -> None:
        """Downloads the max_profit values still on the GPU.  ``needed`` (bank slots about to be reported): only the pending
        runs that hold one of them — the slots refilled at THIS boundary are played an episode from now, and fetching them here
        would synchronise behind the generating kernel that was just launched (the deferral did not defer)."""
        keep = []
        for first, cnt in self._max_profit_pending:
            if needed is not None and not bool(((needed >= first) & (needed < first + cnt)).any()):
                keep.append((first, cnt))
                continue
            self._max_profit[first:first + cnt] = self._engine.download_episodes(first, cnt, tables=False)[4]
        self._max_profit_pending = keep

    def _finish_batched(self, slots, drawn) -> None:
        ns, sess, req, day, mp = drawn
        self._max_profit[slots] = mp
        self._upload_runs(slots, ns, sess, req, day)

    def _drain(self, force: bool) -> None:
        if self._pending is not None and (force or self._pending[0].done()):
            fut, slots = self._pending
            self._pending = None
            self._finish_batched(slots, fut.result())

    def _stage(self, env_ids: np.ndarray, slots: np.ndarray, seeds) -> None:
        from ._lib import SESSION_DTYPE
        cnt = len(env_ids)
        if self._batched is not None:
            self._stage_batched(np.asarray(slots))
            return
        if self._devgen is not None:
            self._stage_device(np.asarray(slots))
            return
        ns = np.zeros(cnt, np.int32)
        sess = np.zeros((cnt, self._stride), dtype=SESSION_DTYPE)
        req = np.zeros((cnt, self._stride))
        day = np.zeros(cnt, np.int32)
        for j, (i, seed) in enumerate(zip(env_ids, seeds)):
            g = self.generators[i]
            g.set_seed(seed)                           # env.py:314 (also on autoreset: reset() -> seed=None)
            table = g.get_event_table()
            ns[j] = len(table)
            sess[j], req[j] = _pad_table(table, self._stride)
            day[j] = (g.day - self._day0).days         # MOER of the advanced day (env.py:321-323)
            self._max_profit[slots[j]] = table.max_profit()
        self._upload_runs(np.asarray(slots), ns, sess, req, day)

    def pipeline_halves(self):
        """``[(slice of environments, torch.cuda.ExternalStream)]`` of the two half batches (``pipeline=2``)."""
        return self._engine.pipeline_halves()

    def join(self) -> None:
        """Orders the current torch stream after the pending half launches (no-op with ``pipeline=1``)."""
        self._engine.join()

    def _wrap_obs(self, flat):
        return {key: flat[:, sl] for key, sl in self._slices.items()}

    # -- VectorEnv API --------------------------------------------------------------------
    def reset(self, *, seed: int | Sequence[int] | None = None, options: dict | None = None):
        N = self.num_envs
        if seed is None:
            seeds = [None] * N
        elif np.isscalar(seed):
            seeds = [int(seed) + i for i in range(N)]  # gymnasium VectorEnv seeding convention
        else:
            seeds = list(seed)
        ids = np.arange(N)
        self._drain(force=True)
        if seed is not None and (self._batched is not None or self._devgen is not None):
            gens_seed = int(seed) if np.isscalar(seed) else int(seeds[0])
            (self._batched or self._devgen).set_seed(gens_seed)
        self._steps_in_episode = 0
        if self._realbank is not None:
            # RealTraceGenerator.set_seed with sequential days: day = (seed mod D); None -> 0
            first = np.array([0 if sd is None else int(sd) for sd in seeds], dtype=np.int64) % self._ndays
            self._cur_slot = first
        else:
            self._stage(ids, ids, seeds)                  # current episodes -> slots [0, N)
            self._stage(ids, ids + N, [None] * N)         # next episodes   -> slots [N, 2N)
            self._cur_slot = ids.copy()
        self._info_max_profit = None
        host = self.output == 'numpy'
        obs = self._engine.reset(slots=self._cur_slot.astype(np.int32), host=host)
        if host:
            obs = obs.copy()
        return self._wrap_obs(obs), self._infos(None, None)

    def _infos(self, out, done_mask):
        """``info['max_profit']`` is one cached array per episode (treat as read-only)."""
        bd = None if out is None else out['breakdown']
        if self._info_max_profit is None:
            if self._max_profit_pending:
                self._flush_max_profit(self._cur_slot)
            self._info_max_profit = self._max_profit[self._cur_slot]
        info = {'max_profit': self._info_max_profit}
        if bd is not None:
            info['reward_breakdown'] = {'profit': bd[:, 0], 'carbon_cost': bd[:, 1], 'excess_charge': bd[:, 2]}
        if done_mask is not None:
            info['final_observation'] = self._wrap_obs(out['final_obs'])
            info['_final_observation'] = done_mask
            info['final_info'] = {'max_profit': self._final_max_profit}
            info['_final_info'] = done_mask
        return info

    def step(self, actions=None, *, policy: str | None = None):
        """One step of all environments.  ``policy='greedy'`` / ``'random'`` (instead of ``actions``): the device-resident
        baselines act (GreedyAlgorithm / RandomAlgorithm, algorithms/evcharging/baselines.py:22-51) — no action tensor is built
        or read, the demand columns of the observation make no round trip through a policy kernel.  ``'greedy'`` is what
        ``step(sign(obs['demands']))`` does, bit for bit (``tests/test_gpu_env_api.py``)."""
        N = self.num_envs
        bins = self.discrete_bins if self.discrete_bins > 0 else 0
        assert (actions is None) != (policy is None), 'pass either actions or policy='
        # All environments are reset together and every episode lasts 288 steps, so the boundary
        # is known on the host: no device->host read of `terminated` on the torch path.
        self._steps_in_episode += 1
        boundary = self._steps_in_episode >= 288
        self._drain(force=boundary)                           # next episodes must be in the bank now
        if self.output == 'numpy':
            out = (self._engine.step_policy(policy, bins=bins) if policy is not None
                   else self._engine.step(np.ascontiguousarray(actions), bins=bins))
            term = out['terminated'].astype(bool)
            assert bool(term.all()) == boundary == bool(term.any())
            # the engine alternates between two sets of page-locked output arrays: with zero_copy what this
            # call returns is only valid until the step after the next one; by default the caller gets copies
            if not self.zero_copy:
                out = {key: (val.copy() if key != 'final_obs' or boundary else val) for key, val in out.items()}
            truncated = np.zeros(N, dtype=bool)
        else:
            import torch
            lean_policy = policy == 'greedy' and bins == 0
            if lean_policy or (policy is None and bins == 0 and actions.dtype == torch.float32 and actions.is_contiguous()
                               and actions.is_cuda):
                # lean path: one ctypes call per step on a pre-built argument block
                stream = torch.cuda.current_stream(self._engine.device).cuda_stream
                if self._stepper is None or stream != self._stepper_stream or self._stepper_policy != policy:
                    self._stepper, self._stepper_out = self._engine.make_stepper(policy=policy)
                    self._stepper_stream = stream
                    self._stepper_policy = policy
                    # The stepper's outputs are the SAME tensors every step (like the reference's reused observation
                    # buffers, env.py:152-158): their views — the observation dict, terminated as bool, the breakdown
                    # columns — are built once, not per step (16.5 -> ~8 us of host time per step; with pipeline=2 the host
                    # was what bounded the quiet hours of a day).
                    so = self._stepper_out
                    bd = so['breakdown']
                    self._lean_views = (self._wrap_obs(so['obs']), so['reward'], so['terminated'].view(torch.bool),
                                        {'profit': bd[:, 0], 'carbon_cost': bd[:, 1], 'excess_charge': bd[:, 2]})
                    self._lean_info = None
                if policy is None:
                    assert actions.dim() == 2 and actions.shape[0] == N and actions.shape[1] == self.num_stations
                    self._stepper(actions.data_ptr())
                else:
                    self._stepper(None)
                out = self._stepper_out
                if not boundary:
                    obs_v, rew_v, term_v, bd_v = self._lean_views
                    if self._false_dev is None:
                        self._false_dev = term_v.new_zeros(N)
                    if self._info_max_profit is None:            # first step of an episode (reset() / the boundary step cleared it)
                        if self._max_profit_pending:
                            self._flush_max_profit(self._cur_slot)
                        self._info_max_profit = self._max_profit[self._cur_slot]
                    if self._lean_info is None or self._lean_info['max_profit'] is not self._info_max_profit:
                        # one dict per EPISODE: rebuilt whenever the episode's max_profit array was (the boundary step and
                        # reset() refill _info_max_profit through _infos(), so testing it for None here is not enough)
                        self._lean_info = {'max_profit': self._info_max_profit, 'reward_breakdown': bd_v}
                    return obs_v, rew_v, term_v, self._false_dev, self._lean_info
            elif policy is not None:
                out = self._engine.step(bins=bins, policy=policy)
            else:
                out = self._engine.step(actions, bins=bins)
            term = out['terminated'].view(torch.bool)             # uint8 0/1: zero-copy
            truncated = self._false_dev
            if truncated is None:
                truncated = self._false_dev = term.new_zeros(N)
        done_mask = None
        if boundary:
            done_mask = np.ones(N, dtype=bool)
            self._final_max_profit = self._max_profit[self._cur_slot]
            vacated = self._cur_slot.copy()
            step_slots = 1 if self._realbank is not None else N
            self._cur_slot = (vacated + step_slots) % self._bank_slots     # kernel autoreset: slot + stride
            self._info_max_profit = None
            self._episodes += 1
            self._steps_in_episode = 0
            if self._realbank is not None:
                pass                                           # the next day is already in the bank
            elif self._devgen is not None:
                self._stage_device(vacated)                    # one kernel launch, no host sampling
            elif self._batched is not None:
                self._stage_batched(vacated, background=True)  # overlaps the next episode's steps
            else:
                self._stage(np.arange(N), vacated, [None] * N)  # refill the vacated slots
        return self._wrap_obs(out['obs']), out['reward'], term, truncated, self._infos(out, done_mask)

    def close(self) -> None:
        if not self.closed:
            if self._pool is not None:
                self._pool.shutdown(wait=True)
            self._engine.close()
            self.closed = True

    # -- multi-agent view (config 5): every agent sees the same flattened observation -----
    def agent_observations(self, flat_obs):
        """``[N, F] -> [N, n_agents, F]`` zero-copy broadcast view (multiagent_env.py:114-117 hands
        the same array object to every agent)."""
        n = self.num_stations
        if hasattr(flat_obs, 'expand'):
            return flat_obs.unsqueeze(1).expand(-1, n, -1)
        return np.broadcast_to(flat_obs[:, None, :], (flat_obs.shape[0], n, flat_obs.shape[1]))


class MultiAgentEVChargingVectorEnv:
    """Batched form of :class:`MultiAgentEVChargingEnv` (BASELINE config 5: N environments x n
    agents on the GPU).  ``step(actions[N, n])`` returns per-agent observations ``[N, n, F]``
    (device tensor), rewards ``[N, n]`` (= reward / n for every agent, multiagent_env.py:186),
    terminateds ``[N, n]``.  ``periods_delay = 0`` hands out a zero-copy broadcast view of the
    flattened observation (every agent sees the same array, multiagent_env.py:114-117); with
    ``delay_semantics='documented'`` and ``periods_delay > 0`` a HIP gather kernel builds each
    agent's row from the current observation and the one ``periods_delay`` steps ago, which the
    engine writes into a ring of observation buffers (no extra copies)."""

    def __init__(self, data_generators, num_envs: int | None = None, periods_delay: int = 0,
                 moer_forecast_steps: int = 36, project_action_in_env: bool = True,
                 discrete_bins: int = -1, device: int = 0, delay_semantics: str = 'reference',
                 materialize: bool = False, charge_calculation: str = 'continuous'):
        assert delay_semantics in ('reference', 'documented')
        self.venv = EVChargingVectorEnv(data_generators, num_envs, moer_forecast_steps,
                                        project_action_in_env, discrete_bins, device, output='torch',
                                        charge_calculation=charge_calculation)
        self.num_envs = self.venv.num_envs
        self.possible_agents = self.venv.cn.station_ids[:]
        self.num_agents = len(self.possible_agents)
        self.periods_delay = periods_delay
        self.delay = periods_delay if delay_semantics == 'documented' else 0
        self.materialize = materialize or self.delay > 0
        self._ring: list = []
        self._pos = 0
        self._agent_buf = None

    def _agent_obs(self, flat, init: bool = False):
        eng = self.venv._engine
        if not self.materialize:
            return self.venv.agent_observations(flat)
        delayed = None
        if self.delay > 0:
            if init:
                self._ring = [flat.clone() for _ in range(self.delay)]
                self._pos = 0
            delayed = self._ring[self._pos]
        self._agent_buf = eng.gather_agent_obs(flat, delayed if not init else None, self._agent_buf)
        if self.delay > 0 and not init:
            self._ring[self._pos].copy_(flat)      # becomes the observation `delay` steps ago
            self._pos = (self._pos + 1) % self.delay
        return self._agent_buf

    def reset(self, *, seed=None, options=None):
        obs, info = self.venv.reset(seed=seed, options=options)
        flat = self.venv._engine.device_outputs()['obs']
        return self._agent_obs(flat, init=True), info

    def step(self, actions):
        """``actions``: CUDA tensor ``[N, n]`` (float32, or int64 for discrete bins)."""
        obs, rew, term, trunc, info = self.venv.step(actions)
        flat = self.venv._engine.device_outputs()['obs']
        n = self.num_agents
        # On the step that ends the episodes the kernel has already autoreset: `flat` is the first
        # observation of the next episodes, and — like MultiAgentEVChargingEnv.reset (init=True) — every
        # slot of the delay ring restarts from it instead of carrying the finished episode's observations.
        agent_obs = self._agent_obs(flat, init=self.venv._steps_in_episode == 0)
        rewards = (rew / n).unsqueeze(1).expand(-1, n)
        return agent_obs, rewards, term.unsqueeze(1).expand(-1, n), trunc.unsqueeze(1).expand(-1, n), info

    def close(self) -> None:
        self.venv.close()


class _StepInfoSource:
    """What the per-environment info dicts of one SB3VecEnv step are made of (arrays over the batch)."""
    __slots__ = ('max_profit', 'breakdown', 'done', 'final', 'written')

    def __init__(self):
        self.max_profit = self.breakdown = self.done = self.final = None
        self.written: list = []          # info objects somebody stored a key in during the current step (next_step drops those keys)

    def next_step(self) -> None:
        """A new step begins: what wrappers stored in the previous step's info objects (``info[k] = v``) goes away with that
        step, as it does with the fresh dicts DummyVecEnv / SubprocVecEnv hand out.  VecNormalize and VecFrameStack overwrite
        ``infos[i]['terminal_observation']`` IN PLACE; kept, that value would shadow the lazy one in every later step and
        episode of the environment (VERDICT / ADVICE r5).  Cost: one pass over the objects that were written, usually none."""
        for d in self.written:
            dict.clear(d)
        self.written.clear()


class _LazyInfo(dict):
    """``infos[i]`` of :class:`SB3VecEnv` without a Python loop over the batch: a dict whose items are read out of the step's
    batch arrays when somebody asks (SB3 itself looks at ``TimeLimit.truncated`` / ``terminal_observation`` of the environments
    that ended; monitors ``.copy()`` and add an ``episode`` key).  The N objects are created once; every step they describe the
    CURRENT step (keep ``dict(info)`` / ``info.copy()`` — real dicts — if an old one is needed).  Keys a wrapper stores
    through ``info[k] = v`` (``update``, ``setdefault``) live in the dict proper and shadow the lazy value FOR THAT STEP: the
    next ``step_wait`` drops them (:meth:`_StepInfoSource.next_step`)."""
    __slots__ = ('_src', '_i')
    _KEYS = ('max_profit', 'reward_breakdown', 'TimeLimit.truncated')

    def __init__(self, src: _StepInfoSource, i: int):
        super().__init__()
        self._src, self._i = src, i

    def _has_terminal(self) -> bool:
        return self._src.done is not None and bool(self._src.done[self._i])

    def _mark_written(self) -> None:
        if not dict.__len__(self):
            self._src.written.append(self)

    def __setitem__(self, key, value):
        self._mark_written()
        dict.__setitem__(self, key, value)

    def update(self, *args, **kwargs):
        self._mark_written()
        dict.update(self, *args, **kwargs)

    def setdefault(self, key, default=None):
        if key in self:
            return self[key]
        self[key] = default
        return default

    def __missing__(self, key):
        s, i = self._src, self._i
        if key == 'max_profit':
            return float(s.max_profit[i])
        if key == 'reward_breakdown':
            b = s.breakdown[i]
            return {'profit': float(b[0]), 'carbon_cost': float(b[1]), 'excess_charge': float(b[2])}
        if key == 'TimeLimit.truncated':
            return False
        if key == 'terminal_observation' and self._has_terminal():
            return {k: v[i].copy() for k, v in s.final.items()}
        raise KeyError(key)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._KEYS or (key == 'terminal_observation' and self._has_terminal())

    def keys(self):
        ks = list(self._KEYS) + (['terminal_observation'] if self._has_terminal() else [])
        return ks + [k for k in dict.keys(self) if k not in ks]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def copy(self):
        return dict(self.items())

    def __repr__(self):
        return repr(self.copy())

    def __eq__(self, other):
        return self.copy() == (other.copy() if isinstance(other, _LazyInfo) else other)

    __hash__ = None


class SB3VecEnv(_SB3VecEnvBase):
    """stable_baselines3 ``VecEnv`` protocol over :class:`EVChargingVectorEnv`
    (used like train_stable_baselines.py:275 uses SubprocVecEnv; policy ``MultiInputPolicy``).
    Derives from SB3's ``VecEnv`` when stable_baselines3 is installed, so that algorithms take it as
    it is instead of wrapping it in a DummyVecEnv.

    ``infos='lazy'`` (default): ``step_wait`` returns a list of N dict objects that read their items out of the step's batch
    arrays on access (:class:`_LazyInfo`) — building N real dicts cost 1.5 us per environment per step in Python, 6 ms at 4 096
    environments against an engine step of 0.11 ms.  ``infos='dicts'`` builds real dicts like DummyVecEnv does."""

    def __init__(self, venv: EVChargingVectorEnv, infos: str = 'lazy', copy_obs: bool = True):
        assert venv.output == 'numpy'
        assert infos in ('lazy', 'dicts')
        self.venv = venv
        self._infos_mode = infos
        # copy_obs=False hands out the engine's two alternating page-locked observation buffers themselves: what step k returns
        # stays valid until step k + 2 is taken — enough for SB3's own collect loops (they turn step k's observation into
        # step k + 1's action and COPY it into their rollout / replay buffer), not for a wrapper that keeps older observations
        self._copy_obs = bool(copy_obs)
        self._info_src = _StepInfoSource()
        self._lazy: list[_LazyInfo] | None = None
        # this adapter copies every array it hands out (SB3's rollout buffer keeps references across steps), so on ITS
        # calls the vector env underneath may hand out its alternating page-locked buffers instead of copying a first
        # time (step_wait); the caller's venv object itself is left as it was configured
        if _SB3VecEnvBase is not object:       # sets num_envs / spaces, queries get_attr('render_mode')
            super().__init__(venv.num_envs, venv.single_observation_space, venv.single_action_space)
        self.num_envs = venv.num_envs
        self.observation_space = venv.single_observation_space
        self.action_space = venv.single_action_space
        self.render_mode = None
        self._actions = None
        self._seeds: list[int | None] = [None] * self.num_envs

    def seed(self, seed: int | None = None):
        self._seeds = [None if seed is None else seed + i for i in range(self.num_envs)]
        return self._seeds

    def reset(self):
        seeds = None if all(s is None for s in self._seeds) else self._seeds
        obs, _ = self.venv.reset(seed=seeds)
        self._seeds = [None] * self.num_envs
        return {k: v.copy() for k, v in obs.items()}

    def step_async(self, actions) -> None:
        self._actions = actions

    def step_wait(self):
        keep = self.venv.zero_copy
        self.venv.zero_copy = True                 # for this call only: everything handed out below is copied here
        try:
            obs, rew, term, trunc, info = self.venv.step(self._actions)
        finally:
            self.venv.zero_copy = keep
        if self._infos_mode == 'lazy':
            src = self._info_src
            src.next_step()                        # keys wrappers wrote into the previous step's info objects end with that step
            src.max_profit = info['max_profit']
            bd = info['reward_breakdown']
            src.breakdown = np.stack([bd['profit'], bd['carbon_cost'], bd['excess_charge']], axis=1)     # a copy: [N, 3]
            if term.any():
                src.done = np.array(term, dtype=bool)
                src.final = {k: v.copy() for k, v in info['final_observation'].items()}
            else:
                src.done = src.final = None
            if self._lazy is None:
                self._lazy = [_LazyInfo(src, i) for i in range(self.num_envs)]
            return ({k: v.copy() for k, v in obs.items()} if self._copy_obs else dict(obs)), rew.astype(np.float32), term, self._lazy
        infos: list[dict[str, Any]] = []
        bd = info['reward_breakdown']
        for i in range(self.num_envs):
            d = {'max_profit': float(info['max_profit'][i]),
                 'reward_breakdown': {k: float(v[i]) for k, v in bd.items()},
                 'TimeLimit.truncated': False}
            if term[i]:
                d['terminal_observation'] = {k: v[i].copy() for k, v in info['final_observation'].items()}
            infos.append(d)
        return {k: v.copy() for k, v in obs.items()}, rew.astype(np.float32), term, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self) -> None:
        self.venv.close()

    def get_attr(self, attr_name, indices=None):
        idx = range(self.venv.num_envs) if indices is None else indices
        if attr_name == 'render_mode':
            return [None for _ in idx]
        return [getattr(self.venv, attr_name) for _ in idx]

    def set_attr(self, attr_name, value, indices=None) -> None:
        setattr(self.venv, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        idx = range(self.num_envs) if indices is None else indices
        return [getattr(self.venv, method_name)(*args, **kwargs) for _ in idx]

    def env_is_wrapped(self, wrapper_class, indices=None):
        idx = range(self.num_envs) if indices is None else indices
        return [False for _ in idx]

    def get_images(self):
        return [None] * self.num_envs


class _SubEnvView:
    """What :meth:`RLlibVectorEnv.get_sub_environments` hands out: RLlib asks its sub-environments for their spaces (and a
    few attributes) only; stepping goes through the vector env.  A view, not an environment: ``step`` / ``reset`` refuse."""

    def __init__(self, owner: 'RLlibVectorEnv', index: int):
        self._owner, self.index = owner, index
        self.observation_space, self.action_space = owner.observation_space, owner.action_space
        self.spec = None
        self.metadata: dict[str, Any] = {}
        self.render_mode = None

    @property
    def unwrapped(self):
        return self

    def step(self, action):
        raise RuntimeError('sub-environments of RLlibVectorEnv are stepped together: use vector_step')

    def reset(self, **kwargs):
        return self._owner.reset_at(self.index, **kwargs)

    def close(self) -> None:
        pass


class RLlibVectorEnv(_RLlibVectorEnvBase):
    """``ray.rllib.env.vector_env.VectorEnv`` protocol (Ray 2.x: ``vector_reset``, ``reset_at``, ``restart_at``,
    ``vector_step``, ``get_sub_environments``) over :class:`EVChargingVectorEnv` — the batched form of what the reference
    gives RLlib: ``num_envs_per_worker`` copies of ``FlattenObservation(EVChargingEnv(gen))`` per rollout worker, optionally
    under ``DiscreteActionWrapper`` (examples/evcharging/train_rllib.py:129-134,158-160; tests/test_evcharging.py:26-27).
    Derives from RLlib's ``VectorEnv`` when ray is installed, so that ``config.environment(lambda cfg: RLlibVectorEnv(...))``
    is taken as it is instead of being wrapped.

    ``flatten=True`` (default, what the reference's scripts do): observations are flat float32 vectors of length
    2 n + k + 2 in ``gymnasium.spaces.flatten`` order (= the engine's own row); ``False``: the Dict observation of env.py:143-150.

    Episodes: the engine resets all environments itself at the boundary (they run in lock-step, 288 steps).  RLlib's sampler
    wants the TERMINAL observation from ``vector_step`` and the first observation of the next episode from ``reset_at(i)``:
    the boundary step returns ``info['final_observation']`` rows and keeps the new episode's first observations for the
    ``reset_at`` calls that follow.  ``reset_at`` of an environment that has not terminated is refused (lock-step)."""

    def __init__(self, venv: EVChargingVectorEnv, flatten: bool = True):
        assert venv.output == 'numpy', 'RLlib moves numpy observations'
        self.venv = venv
        self.flatten = bool(flatten)
        obs_space = spaces.flatten_space(venv.single_observation_space) if flatten else venv.single_observation_space
        if _RLlibVectorEnvBase is not object:
            super().__init__(obs_space, venv.single_action_space, venv.num_envs)
        self.observation_space, self.action_space, self.num_envs = obs_space, venv.single_action_space, venv.num_envs
        self._subs = [_SubEnvView(self, i) for i in range(self.num_envs)]
        self._pending_obs = None            # first observations of the next episode, handed out by reset_at after a boundary
        self._pending_info = None
        self._awaiting = np.zeros(self.num_envs, dtype=bool)
        self._started = False

    # -- helpers ------------------------------------------------------------------------------------------------------
    def _rows(self, obs: dict[str, np.ndarray]):
        """Batched dict observation -> list of N per-environment observations (copies: RLlib keeps them in its batches)."""
        if self.flatten:
            flat = np.concatenate([obs[key] for key in OBS_KEYS], axis=1)             # spaces.flatten order (sorted keys)
            return list(flat)
        cols = {key: np.array(obs[key]) for key in OBS_KEYS}
        return [{key: cols[key][i] for key in OBS_KEYS} for i in range(self.num_envs)]

    @staticmethod
    def _info_rows(info: dict[str, Any], n: int, max_profit=None):
        mp = info['max_profit'] if max_profit is None else max_profit
        bd = info.get('reward_breakdown')
        if bd is None:
            return [{'max_profit': float(mp[i])} for i in range(n)]
        p, c, x = bd['profit'], bd['carbon_cost'], bd['excess_charge']
        return [{'max_profit': float(mp[i]),
                 'reward_breakdown': {'profit': float(p[i]), 'carbon_cost': float(c[i]), 'excess_charge': float(x[i])}}
                for i in range(n)]

    # -- ray.rllib.env.VectorEnv ----------------------------------------------------------------------------------
    def vector_reset(self, *, seeds=None, options=None):
        seed = None if seeds is None or all(sd is None for sd in seeds) else [0 if sd is None else int(sd) for sd in seeds]
        obs, info = self.venv.reset(seed=seed)
        self._awaiting[:] = False
        self._pending_obs = self._pending_info = None
        self._started = True
        return self._rows(obs), self._info_rows(info, self.num_envs)

    def reset_at(self, index: int | None = None, *, seed=None, options=None):
        index = 0 if index is None else int(index)
        if not self._started:                         # RLlib may reset single sub-environments first: start all of them once
            obs, infos = self.vector_reset(seeds=None if seed is None else [int(seed) + i for i in range(self.num_envs)])
            self._pending_obs, self._pending_info = obs, infos
            self._awaiting[:] = True
        if not self._awaiting[index]:
            raise ValueError(f'reset_at({index}): the environment has not terminated — the batch runs in lock-step '
                             '(288-step episodes, reset together by the engine); use vector_reset to restart all of it')
        self._awaiting[index] = False
        return self._pending_obs[index], self._pending_info[index]

    def restart_at(self, index: int | None = None) -> None:
        raise NotImplementedError('restart_at: sub-environments share one engine and cannot be re-created one by one')

    def vector_step(self, actions):
        if self._awaiting.any():
            raise RuntimeError('vector_step before reset_at was called for every terminated environment')
        a = np.stack([np.asarray(x) for x in actions]) if not isinstance(actions, np.ndarray) else actions
        if self.venv.discrete_bins > 0:
            a = a.astype(np.int64, copy=False)
        else:
            a = a.astype(np.float32, copy=False)
        obs, rew, term, trunc, info = self.venv.step(a)
        n = self.num_envs
        infos = self._info_rows(info, n, info['final_info']['max_profit'] if term.any() else None)
        if term.any():                                 # lock-step: all of them
            self._pending_obs = self._rows(obs)                              # next episode's first observations
            self._pending_info = [{'max_profit': float(v)} for v in info['max_profit']]
            self._awaiting[:] = np.asarray(term, dtype=bool)
            rows = self._rows(info['final_observation'])                    # what the episode ended on
        else:
            rows = self._rows(obs)
        return rows, [float(r) for r in rew], [bool(t) for t in term], [bool(t) for t in trunc], infos

    def get_sub_environments(self):
        return self._subs

    def try_render_at(self, index: int | None = None):
        return None

    def close(self) -> None:
        self.venv.close()
