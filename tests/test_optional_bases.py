"""With gymnasium / pettingzoo / stable_baselines3 / ray importable the public classes derive from the reference's bases
(env.py:20, wrappers.py:13, multiagent_env.py:18; SB3VecEnv: stable_baselines3's VecEnv; RLlibVectorEnv: ray.rllib's VectorEnv) and the env id of sustaingym/__init__.py:3-7 is
registered.  Neither package exists in the build image, so the check runs in a subprocess against
structural stand-ins of the two packages (just the class skeletons gymnasium 0.28 defines)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAKE_GYMNASIUM = '''
class Env:
    metadata = {}

class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self._action_space = None
        self._observation_space = None
    @property
    def action_space(self):
        return self.env.action_space if self._action_space is None else self._action_space
    @action_space.setter
    def action_space(self, space):
        self._action_space = space
    @property
    def observation_space(self):
        return self.env.observation_space if self._observation_space is None else self._observation_space
    @observation_space.setter
    def observation_space(self, space):
        self._observation_space = space

class ActionWrapper(Wrapper):
    def step(self, action):
        return self.env.step(self.action(action))
'''

CHECK = '''
import numpy as np
import gymnasium, pettingzoo
from gymnasium.envs.registration import registry
import sustaingym_amd
from sustaingym_amd import envs, spaces
assert issubclass(envs.EVChargingEnv, gymnasium.Env)
assert issubclass(envs.DiscreteActionWrapper, gymnasium.ActionWrapper)
assert issubclass(envs.MultiAgentEVChargingEnv, pettingzoo.ParallelEnv)
import gymnasium.vector
assert issubclass(envs.EVChargingVectorEnv, gymnasium.vector.VectorEnv)
assert registry['sustaingym/EVCharging-v0'] == 'sustaingym_amd.envs:EVChargingEnv'
assert 'sustaingym_amd/EVCharging-v0' in registry

class Dummy(gymnasium.Env):
    action_space = spaces.Box(0.0, 1.0, shape=(54,))
    observation_space = 'obs-space'
    marker = 'forwarded'
    def step(self, action):
        self.last = action
        return 'o', 0.0, False, False, {}
    def reset(self, **kw):
        return 'o', {}
    def close(self):
        self.closed = True

d = Dummy()
w = envs.DiscreteActionWrapper(d, bins=5)
assert isinstance(w, gymnasium.Env)
assert isinstance(w.action_space, spaces.MultiDiscrete) and w.observation_space == 'obs-space'
w.step(np.full(54, 4))
assert d.last.dtype == np.float32 and np.all(d.last == 1.0)
w.step(np.full(54, 1))
assert np.all(d.last == np.float32(0.25))
assert w.marker == 'forwarded' and w.reset() == ('o', {})
w.close()
assert d.closed

import stable_baselines3.common.vec_env as sb3
assert issubclass(envs.SB3VecEnv, sb3.VecEnv)
class DummyVenv:
    output = 'numpy'
    num_envs = 3
    single_observation_space = 'single-obs'
    single_action_space = 'single-act'
v = envs.SB3VecEnv(DummyVenv())
assert v.num_envs == 3 and v.observation_space == 'single-obs' and v.action_space == 'single-act'
assert v.render_mode is None and v.base_init_ran

from ray.rllib.env.vector_env import VectorEnv as RayVectorEnv
assert issubclass(envs.RLlibVectorEnv, RayVectorEnv)
class DummyVenv2(DummyVenv):
    single_observation_space = envs.make_observation_space(4, 2, 100.0)
    discrete_bins = -1
r = envs.RLlibVectorEnv(DummyVenv2())
assert r.ray_init == (r.observation_space, 'single-act', 3) and r.observation_space.shape == (12,)
assert len(r.get_sub_environments()) == 3
print('OK')
'''


def test_reference_bases_when_packages_exist(tmp_path):
    g = tmp_path / 'gymnasium'
    (g / 'envs').mkdir(parents=True)
    (g / '__init__.py').write_text(FAKE_GYMNASIUM)
    (g / 'spaces.py').write_text("raise ImportError('stand-in without spaces: sustaingym_amd.spaces falls back')\n")
    (g / 'envs' / '__init__.py').write_text('')
    (g / 'vector').mkdir()
    (g / 'vector' / '__init__.py').write_text('class VectorEnv:\n    pass\n')
    (g / 'vector' / 'utils.py').write_text('def batch_space(space, n):\n    return (space, n)\n')
    (g / 'envs' / 'registration.py').write_text(textwrap.dedent('''
        registry = {}
        def register(id, entry_point, **kwargs):
            registry[id] = entry_point
    '''))
    sb = tmp_path / 'stable_baselines3' / 'common'
    sb.mkdir(parents=True)
    (tmp_path / 'stable_baselines3' / '__init__.py').write_text('')
    (sb / '__init__.py').write_text('')
    (sb / 'vec_env.py').write_text(textwrap.dedent('''
        class VecEnv:
            def __init__(self, num_envs, observation_space, action_space):
                self.num_envs, self.observation_space, self.action_space = num_envs, observation_space, action_space
                modes = self.get_attr('render_mode')          # as stable_baselines3 >= 2.0 does
                assert len(modes) == num_envs
                self.render_mode = modes[0]
                self.base_init_ran = True
    '''))
    rv = tmp_path / 'ray' / 'rllib' / 'env'
    rv.mkdir(parents=True)
    for d in (tmp_path / 'ray', tmp_path / 'ray' / 'rllib', rv):
        (d / '__init__.py').write_text('')
    (rv / 'vector_env.py').write_text(textwrap.dedent('''
        class VectorEnv:                     # ray.rllib.env.vector_env.VectorEnv.__init__ (Ray 2.x)
            def __init__(self, observation_space, action_space, num_envs):
                self.observation_space, self.action_space, self.num_envs = observation_space, action_space, num_envs
                self.ray_init = (observation_space, action_space, num_envs)
    '''))
    p = tmp_path / 'pettingzoo'
    p.mkdir()
    (p / '__init__.py').write_text('class ParallelEnv:\n    pass\n')
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), ROOT]))
    out = subprocess.run([sys.executable, '-c', CHECK], env=env, capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0 and out.stdout.strip().endswith('OK'), out.stderr[-2000:]
