"""SB3VecEnv.step per call, lazy infos against real dicts (GPU box): `python tools/sb3_step_cost.py [N]`"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from sustaingym_amd.envs import EVChargingVectorEnv, SB3VecEnv
from sustaingym_amd.event_generation import DeviceGMMTraceGenerator
for N in ([int(sys.argv[1])] if len(sys.argv) > 1 else [64, 1024, 4096, 16384]):
    for mode in ('dicts', 'lazy', 'lazy+nocopy'):
        e = SB3VecEnv(EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2021', seed=0), num_envs=N), infos=mode.split('+')[0], copy_obs='+nocopy' not in mode)
        e.reset()
        a = np.random.default_rng(0).random((N, 54), dtype=np.float32)
        for _ in range(10): e.step(a)
        t0 = time.perf_counter()
        for _ in range(100): e.step(a)
        dt = (time.perf_counter() - t0) / 100
        print(f'N={N} infos={mode}: {dt * 1e6:.0f} us per step -> {N / dt / 1e6:.2f} M env-steps/s')
        e.close()
