# Host-buffer (numpy) path of the vector env: what SB3 / RLlib style callers pay per step.
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sustaingym_amd.envs import EVChargingVectorEnv
from sustaingym_amd.event_generation import DeviceGMMTraceGenerator
for N in (4096, 4096, 16384, 65536):
    venv = EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2021', seed=0), num_envs=N, output='numpy', zero_copy=True)
    venv.reset(seed=0)
    a = np.random.default_rng(0).random((N, 54), dtype=np.float32)
    for _ in range(10): venv.step(a)
    t0 = time.perf_counter()
    for _ in range(100): venv.step(a)
    dt = (time.perf_counter() - t0) / 100
    print(f'N={N}: {dt*1e6:.0f} us/step -> {N/dt/1e6:.1f} M env-steps/s (host buffers: {a.nbytes/1e6:.1f} MB in, {N*146*4/1e6:.1f} MB obs out)')
    venv.close()
