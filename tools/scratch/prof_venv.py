import sys, cProfile, pstats
sys.path.insert(0, '.')
import torch
from sustaingym_amd.envs import EVChargingVectorEnv
from sustaingym_amd.event_generation import DeviceGMMTraceGenerator
N = 65536
venv = EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2019', seed=0), num_envs=N, output='torch')
venv.reset()
acts = torch.rand((N, 54), device='cuda')
for s in range(50): venv.step(acts)
pr = cProfile.Profile(); pr.enable()
for s in range(200): venv.step(acts)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
