"""Baseline controllers that call ``step()`` (reference: sustaingym/algorithms/evcharging/
baselines.py:22-51 and the episode runner sustaingym/algorithms/base.py:38-99).

Only the two arithmetic-free baselines are provided (Greedy, Random); MPC / OfflineOptimal are
cvxpy programs that merely *call* the environment and stay out of scope.  ``GreedyAlgorithm`` also
exists as a device-resident policy (``EVC_ACTION_GREEDY``): the whole episode loop then runs on
the GPU with one C call (``StepEngine.rollout(policy='greedy')``).
"""
from __future__ import annotations

from copy import deepcopy
from typing import Any, Sequence

import numpy as np

from . import spaces

MAX_ACTION = 1      # baselines.py:18
D_MAX_ACTION = 4    # baselines.py:19


class BaseAlgorithm:
    """algorithms/base.py:18-99 (single-agent part)."""

    def __init__(self, env, multiagent: bool = False):
        self.env = env
        self.multiagent = multiagent

    def reset(self) -> None:
        pass

    def get_action(self, observation):
        raise NotImplementedError

    def run(self, seeds: Sequence[int] | int) -> dict[str, list]:
        """base.py:38-99; returns the columns of the reference's DataFrame as a dict of lists."""
        if isinstance(seeds, int):
            seeds = list(range(seeds))
        results: dict[str, list] = {}
        for seed in seeds:
            results.setdefault('seed', []).append(seed)
            ep_return = 0.0
            obs, _ = self.env.reset(seed=seed)
            self.reset()
            done = False
            info: dict[str, Any] = {}
            while not done:
                action = self.get_action(obs)
                obs, reward, terminated, truncated, info = self.env.step(action)
                if self.multiagent:
                    reward = sum(reward.values())
                    done = any(terminated.values()) or any(truncated.values())
                else:
                    done = terminated or truncated
                ep_return += reward
            results.setdefault('return', []).append(ep_return)
            if self.multiagent:
                info = info[list(info.keys())[0]]
            for key, value in info.items():
                results.setdefault(key, []).append(deepcopy(value))
        return results


class GreedyAlgorithm(BaseAlgorithm):
    """baselines.py:22-35: full rate wherever the observed demand is non-zero."""

    def __init__(self, env):
        super().__init__(env, multiagent=False)
        self.continuous_action_space = isinstance(env.action_space, spaces.Box)
        self.max_action = MAX_ACTION if self.continuous_action_space else D_MAX_ACTION

    def get_action(self, observation):
        dtype = np.float32 if self.continuous_action_space else np.int64
        return np.where(observation['demands'] > 0, self.max_action, 0).astype(dtype)


class RandomAlgorithm(BaseAlgorithm):
    """baselines.py:38-51."""

    def __init__(self, env, seed: int | None = None):
        super().__init__(env, multiagent=False)
        self.continuous_action_space = isinstance(env.action_space, spaces.Box)
        self.rng = np.random.default_rng(seed)

    def get_action(self, observation):
        if self.continuous_action_space:
            return self.rng.random(size=self.env.num_stations).astype(np.float32)
        return self.rng.choice(D_MAX_ACTION + 1, size=self.env.num_stations).astype(np.int64)
