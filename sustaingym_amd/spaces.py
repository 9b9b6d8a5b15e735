"""Observation / action spaces.

Uses ``gymnasium.spaces`` when gymnasium is installed (so that the environments are drop-in
under StableBaselines3 / RLLib); otherwise a minimal structural stand-in with the same
attributes (``low/high/shape/dtype``, ``sample``, ``contains``, ``Dict.spaces`` with keys in
sorted order like gymnasium) so the engine can be used and tested without gymnasium.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

try:  # pragma: no cover - gymnasium is absent from the build image
    from gymnasium import spaces as _gs
    Box, Dict, Discrete, MultiDiscrete = _gs.Box, _gs.Dict, _gs.Discrete, _gs.MultiDiscrete
    flatten, flatten_space = _gs.flatten, _gs.flatten_space
    HAVE_GYMNASIUM = True
except Exception:  # ModuleNotFoundError in this image
    HAVE_GYMNASIUM = False

    class _Space:
        def __init__(self, shape, dtype, seed=None):
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)
            self._rng = np.random.default_rng(seed)

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

        def __contains__(self, x):
            return self.contains(x)

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            super().__init__(shape, dtype, seed)
            self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()

        def sample(self):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return (x.shape == self.shape and np.can_cast(x.dtype, self.dtype, 'same_kind')
                    and bool(np.all(x >= self.low) and np.all(x <= self.high)))

        def __repr__(self):
            return f'Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})'

    class Discrete(_Space):
        def __init__(self, n, seed=None):
            super().__init__((), np.int64, seed)
            self.n = int(n)

        def sample(self):
            return np.int64(self._rng.integers(self.n))

        def contains(self, x):
            return np.issubdtype(np.asarray(x).dtype, np.integer) and 0 <= int(x) < self.n

        def __repr__(self):
            return f'Discrete({self.n})'

    class MultiDiscrete(_Space):
        def __init__(self, nvec, dtype=np.int64, seed=None):
            self.nvec = np.asarray(nvec, dtype=dtype)
            super().__init__(self.nvec.shape, dtype, seed)

        def sample(self):
            return (self._rng.random(self.nvec.shape) * self.nvec).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))

        def __repr__(self):
            return f'MultiDiscrete({self.nvec})'

    class Dict(_Space):
        def __init__(self, spaces, seed=None):
            # gymnasium sorts the keys of a plain dict (the reference relies on it: the flattened
            # per-agent observation is demands, est_departures, forecasted_moer, prev_moer, timestep)
            self.spaces = OrderedDict(sorted(spaces.items()))
            super().__init__((), np.float64, seed)

        def sample(self):
            return OrderedDict((k, s.sample()) for k, s in self.spaces.items())

        def contains(self, x):
            return (isinstance(x, dict) and set(x) == set(self.spaces)
                    and all(self.spaces[k].contains(v) for k, v in x.items()))

        def keys(self):
            return self.spaces.keys()

        def __getitem__(self, k):
            return self.spaces[k]

        def __repr__(self):
            return 'Dict(' + ', '.join(f'{k}: {v!r}' for k, v in self.spaces.items()) + ')'

    def flatten_space(space):
        if isinstance(space, Dict):
            parts = [flatten_space(s) for s in space.spaces.values()]
            return Box(np.concatenate([p.low for p in parts]), np.concatenate([p.high for p in parts]),
                       dtype=np.result_type(*[p.dtype for p in parts]))
        if isinstance(space, Box):
            return Box(space.low.ravel(), space.high.ravel(), dtype=space.dtype)
        raise NotImplementedError(type(space))

    def flatten(space, x):
        if isinstance(space, Dict):
            return np.concatenate([flatten(s, x[k]) for k, s in space.spaces.items()])
        return np.asarray(x, dtype=space.dtype).ravel()
