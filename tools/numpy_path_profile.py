"""Where a numpy-path step's fixed cost goes (GPU box): `python tools/numpy_path_profile.py [N]`"""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np
from sustaingym_amd.envs import EVChargingVectorEnv
from sustaingym_amd.event_generation import DeviceGMMTraceGenerator
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
venv = EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2021', seed=0), num_envs=N, output='numpy', zero_copy=True)
venv.reset(seed=0)
a = np.random.default_rng(0).random((N, 54), dtype=np.float32)
for _ in range(50): venv.step(a)
t0 = time.perf_counter()
for _ in range(200): venv.step(a)
print(f'N={N}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per step')
eng = venv._engine
t0 = time.perf_counter()
for _ in range(200): eng.step(a)
print(f'  StepEngine.step alone: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us')
pr = cProfile.Profile(); pr.enable()
for _ in range(200): venv.step(a)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
