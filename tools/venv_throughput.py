"""End-to-end rate of EVChargingVectorEnv (torch output) fed by BatchedGMMTraceGenerator:
includes episode sampling at the boundaries (overlapped on a worker thread)."""
import sys, time
sys.path.insert(0, '.')
import torch
from sustaingym_amd.envs import EVChargingVectorEnv
from sustaingym_amd.event_generation import BatchedGMMTraceGenerator

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
site = sys.argv[2] if len(sys.argv) > 2 else 'caltech'
bg = BatchedGMMTraceGenerator(site, 'Summer 2019', seed=0)
t = time.time()
venv = EVChargingVectorEnv(bg, num_envs=N, output='torch')
obs, info = venv.reset()
torch.cuda.synchronize()
print(f'construct+reset {time.time() - t:.2f}s')
acts = torch.rand((N, venv.num_stations), device='cuda')
for ep in range(3):
    t = time.time()
    for s in range(288):
        obs, rew, term, trunc, info = venv.step(acts)
    torch.cuda.synchronize()
    dt = time.time() - t
    print(f'episode {ep}: {dt:.3f}s  {N * 288 / dt / 1e6:.1f} M env-steps/s (pending refill: {venv._pending is not None})')
venv.close()
