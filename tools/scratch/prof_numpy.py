import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sustaingym_amd.envs import EVChargingVectorEnv
from sustaingym_amd.event_generation import DeviceGMMTraceGenerator
N = 4096
venv = EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2021', seed=0), num_envs=N, output='numpy')
venv.reset(seed=0)
a = np.random.default_rng(0).random((N, 54), dtype=np.float32)
for _ in range(10): venv.step(a)
pr = cProfile.Profile(); pr.enable()
for _ in range(100): venv.step(a)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(10)
