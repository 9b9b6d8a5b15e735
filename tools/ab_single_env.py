# BASELINE config 0: one EVChargingEnv (Caltech, DiscreteActionWrapper bins=5) through the Gymnasium API.
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sustaingym_amd import EVChargingEnv, DiscreteActionWrapper, GMMsTraceGenerator, RealTraceGenerator
for name, gen, proj in (('GMM, projection on', GMMsTraceGenerator('caltech', 'Summer 2021'), True),
                        ('real traces, projection off', RealTraceGenerator('caltech', 'Summer 2021'), False)):
    env = DiscreteActionWrapper(EVChargingEnv(gen, project_action_in_env=proj))
    obs, info = env.reset(seed=0)
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 5, (288, 54))
    t0 = time.perf_counter()
    for t in range(288):
        obs, r, term, trunc, info = env.step(acts[t])
    dt = time.perf_counter() - t0
    t1 = time.perf_counter(); env.reset(seed=1); tr = time.perf_counter() - t1
    print(f'{name}: step {dt/288*1e6:.0f} us (host round trip per step), reset {tr*1e3:.1f} ms, terminated={term}')
    env.close()
