import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import torch
from sustaingym_amd.envs import EVChargingVectorEnv
from sustaingym_amd.event_generation import DeviceGMMTraceGenerator, BatchedGMMTraceGenerator
for name, gen, N in (('device', DeviceGMMTraceGenerator('caltech', 'Summer 2019', seed=0), 65536), ('batched-host', BatchedGMMTraceGenerator('caltech', 'Summer 2019', seed=0), 16384)):
    t0 = time.perf_counter(); venv = EVChargingVectorEnv(gen, num_envs=N, output='torch'); t1 = time.perf_counter()
    venv.reset(seed=0); torch.cuda.synchronize(); t2 = time.perf_counter()
    venv.reset(seed=1); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f'{name} N={N}: construct {t1-t0:.3f} s, first reset {t2-t1:.3f} s, second reset {t3-t2:.3f} s')
    if name == 'device':
        pr = cProfile.Profile(); pr.enable(); venv.reset(seed=2); torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(8)
    venv.close()
