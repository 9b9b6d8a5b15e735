import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from sustaingym_amd import EVChargingEnv, DiscreteActionWrapper, GMMsTraceGenerator
for wrap in (True, False):
  for mode in ('0', None):
    if mode is None: os.environ.pop('EVC_HOST_DIRECT_MAX_BYTES', None)
    else: os.environ['EVC_HOST_DIRECT_MAX_BYTES'] = mode
    env = EVChargingEnv(GMMsTraceGenerator('caltech', 'Summer 2021'), project_action_in_env=True)
    if wrap: env = DiscreteActionWrapper(env)
    env.reset(seed=0)
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 5, (288, 54)) if wrap else rng.random((288, 54), dtype=np.float32)
    ts = []
    for t in range(288):
        t0 = time.perf_counter(); env.step(acts[t]); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print(f'wrapper={wrap} direct={"off" if mode else "on"}: mean {ts.mean():.0f} us, median {np.median(ts):.0f}, max {ts.max():.0f} at step {ts.argmax()}, first 3 {ts[:3].round(0)}')
    env.close()
