"""Whole-episode evaluation of arithmetic-free policies for a BATCH of seeds in one engine call.

What the reference does with a Python loop — ``BaseAlgorithm.run(seeds)`` resets the environment
with every seed in turn and steps a ``GreedyAlgorithm`` / ``RandomAlgorithm`` until the episode ends
(sustaingym/algorithms/base.py:38-99, algorithms/evcharging/baselines.py:22-51) — is done here the
other way round: every seed becomes one environment of a batch, the policy lives on the GPU
(``EVC_ACTION_GREEDY`` / ``EVC_ACTION_RANDOM``) and ``evc_rollout`` plays all 288 periods of all
episodes in ONE kernel launch (csrc/evc_rollout.h: the environments' state stays in registers, no
per-period action read or observation write).  The result has the columns of the reference's DataFrame
(``seed``, ``return``, ``max_profit``, ``reward_breakdown``).

MPC / OfflineOptimal (baselines.py:54-223) are cvxpy programs that merely call ``step()``; they stay
out of scope.  Policies that need Python (a trained network, a user's controller) go
through :func:`run_policy`, which drives any environment of this package — ``EVChargingEnv``, a wrapped
one, the multi-agent one — through its own ``reset`` / ``step``.
"""
from __future__ import annotations

from typing import Iterable, Sequence

import numpy as np

from . import _lib
from .hostio import to_host
from .engine import StepEngine
from .event_generation import AbstractTraceGenerator
from .network import site_str_to_site

POLICIES = ('greedy', 'random')
EPISODE_STEPS = _lib.EPISODE_STEPS


def _episode_bank(generator: AbstractTraceGenerator, seeds: Sequence[int], stride: int):
    """One episode per seed, exactly as ``env.reset(seed=seed)`` would create it (env.py:314-323:
    ``set_seed``, event table, then the MOER matrix of the already-advanced day)."""
    ns = np.zeros(len(seeds), np.int32)
    sessions = np.zeros((len(seeds), stride), dtype=_lib.SESSION_DTYPE)
    requested = np.zeros((len(seeds), stride), np.float64)
    moer = np.zeros((len(seeds), _lib.MOER_ROWS, _lib.MOER_COLS), np.float64)
    max_profit = np.zeros(len(seeds), np.float64)
    for i, seed in enumerate(seeds):
        generator.set_seed(int(seed))
        table = generator.get_event_table()
        if len(table) > stride:
            raise ValueError(f'seed {seed}: {len(table)} sessions exceed the engine capacity {stride}')
        ns[i] = len(table)
        sessions[i, :ns[i]] = table.sessions
        requested[i, :ns[i]] = table.requested
        max_profit[i] = table.max_profit()
        moer[i] = generator.get_moer()
    return ns, sessions, requested, moer, max_profit


class PolicyRollout:
    """Plays one episode per seed under a device-resident policy.

    Args:
        data_generator: a trace generator of this package (``RealTraceGenerator``,
            ``GMMsTraceGenerator``): seed ``s`` yields the episode ``EVChargingEnv.reset(seed=s)`` plays.
        policy: ``'greedy'`` — full rate wherever the observed demand is non-zero (baselines.py:32-35) —
            or ``'random'`` — uniform actions (baselines.py:45-51) from the engine's counter-based
            stream (``evc_set_policy_seed``; numpy's sequential ``default_rng`` cannot be drawn by
            thousands of wavefronts, so action values differ from the reference's stream by design).
        discrete_bins: ``> 0`` = the policy acts through ``DiscreteActionWrapper(bins)``
            (wrappers.py:13-45); greedy is unaffected (its maximum action maps to 1.0 either way).
    """

    def __init__(self, data_generator: AbstractTraceGenerator, policy: str = 'greedy',
                 moer_forecast_steps: int = 36, project_action_in_env: bool = True,
                 discrete_bins: int = -1, policy_seed: int = 0, device: int = 0,
                 max_sessions: int = _lib.MAX_SESSIONS, charge_calculation: str = 'continuous'):
        if policy not in POLICIES:
            raise ValueError(f'policy must be one of {POLICIES}')
        self.data_generator = data_generator
        self.policy = policy
        self.cn = site_str_to_site(data_generator.site)
        self.k = int(moer_forecast_steps)
        self.project = bool(project_action_in_env)
        self.bins = int(discrete_bins) if discrete_bins and discrete_bins > 0 else 0
        self.policy_seed = int(policy_seed)
        self.device = int(device)
        self.max_sessions = int(max_sessions)
        self.charge_calculation = charge_calculation

    def run(self, seeds: Iterable[int] | int, env_id_base: int = 0) -> dict[str, np.ndarray | list]:
        """All episodes at once.  Returns ``{'seed', 'return', 'max_profit', 'reward_breakdown',
        'status'}``; ``reward_breakdown`` is a list of dicts like the reference's ``info`` column."""
        import torch
        seeds = list(range(seeds)) if isinstance(seeds, int) else [int(s) for s in seeds]
        if not seeds:
            return {'seed': [], 'return': np.zeros(0), 'max_profit': np.zeros(0), 'reward_breakdown': [],
                    'status': np.zeros(0, np.int32)}
        B = len(seeds)
        ns, sessions, requested, moer, max_profit = _episode_bank(self.data_generator, seeds, self.max_sessions)
        eng = StepEngine(self.cn, B, moer_forecast_steps=self.k, project_action=self.project, autoreset=False,
                         device=self.device, bank_slots=B, max_sessions=self.max_sessions, moer_days=B,
                         charge_calculation=self.charge_calculation)
        try:
            eng.upload_moer(moer)
            eng.upload_episodes(ns, sessions, requested, np.arange(B, dtype=np.int32))
            eng.set_policy_seed(self.policy_seed, env_id_base)
            eng.reset()
            out = eng.rollout(policy=self.policy, steps=EPISODE_STEPS, bins=self.bins)
            torch.cuda.synchronize(self.device)
            assert bool(out['terminated'].all()), 'episodes must end after 288 periods'
            returns = to_host(out['returns']).copy()
            breakdown = to_host(out['breakdown']).copy()
            status = eng.env_scalars()['status']
        finally:
            eng.close()
        return {
            'seed': seeds, 'return': returns, 'max_profit': max_profit,
            'reward_breakdown': [{'profit': float(b[0]), 'carbon_cost': float(b[1]), 'excess_charge': float(b[2])}
                                 for b in breakdown],
            'status': status,
        }

    def run_frame(self, seeds: Iterable[int] | int):
        """``run`` as a pandas DataFrame (the reference's return type)."""
        import pandas as pd
        res = self.run(seeds)
        return pd.DataFrame({k: (list(v) if isinstance(v, np.ndarray) else v) for k, v in res.items()})


def evaluate(data_generator: AbstractTraceGenerator, policy: str, seeds: Iterable[int] | int, **kwargs):
    """Shorthand: ``PolicyRollout(data_generator, policy, **kwargs).run(seeds)``."""
    return PolicyRollout(data_generator, policy, **kwargs).run(seeds)


def run_policy(env, policy, seeds: Iterable[int] | int) -> dict[str, list]:
    """Evaluation of a host-side policy on ANY single environment of this package through its public API: one
    episode per seed, ``policy(observation) -> action`` asked once per period.  For callers of the reference's
    ``BaseAlgorithm.run`` whose controller is Python code (what the device-resident :class:`PolicyRollout` cannot
    take).  Multi-agent environments (dict-of-agents observations) work too: rewards are summed over agents and an
    episode ends when every agent is done.  Returns the columns of :meth:`PolicyRollout.run`."""
    seeds = list(range(seeds)) if isinstance(seeds, int) else [int(s) for s in seeds]
    out: dict[str, list] = {'seed': [], 'return': [], 'max_profit': [], 'reward_breakdown': []}
    for seed in seeds:
        obs, info = env.reset(seed=seed)
        total, finished = 0.0, False
        while not finished:
            obs, reward, terminated, truncated, info = env.step(policy(obs))
            if isinstance(reward, dict):                                  # PettingZoo-parallel style
                total += float(sum(reward.values()))
                finished = all(terminated.values()) or all(truncated.values()) or not terminated
                info = next(iter(info.values())) if info else {}
            else:
                total += float(reward)
                finished = bool(terminated) or bool(truncated)
        out['seed'].append(seed)
        out['return'].append(total)
        out['max_profit'].append(info.get('max_profit'))
        out['reward_breakdown'].append(info.get('reward_breakdown'))
    return out
