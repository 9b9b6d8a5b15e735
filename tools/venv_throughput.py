"""End-to-end rate of EVChargingVectorEnv (torch output) including episode generation at the
boundaries: host sampling (BatchedGMMTraceGenerator, worker thread) vs on-device generation."""
import sys, time
sys.path.insert(0, '.')
import torch
from sustaingym_amd.envs import EVChargingVectorEnv
from sustaingym_amd.event_generation import BatchedGMMTraceGenerator, DeviceGMMTraceGenerator

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
site = sys.argv[2] if len(sys.argv) > 2 else 'caltech'
for name, gen in (('host-batched', BatchedGMMTraceGenerator(site, 'Summer 2019', seed=0)),
                  ('device', DeviceGMMTraceGenerator(site, 'Summer 2019', seed=0))):
    t = time.time()
    venv = EVChargingVectorEnv(gen, num_envs=N, output='torch')
    obs, info = venv.reset()
    torch.cuda.synchronize()
    print(f'[{name}] construct+reset {time.time() - t:.2f}s')
    acts = torch.rand((N, venv.num_stations), device='cuda')
    for ep in range(3):
        t = time.time()
        for s in range(288):
            obs, rew, term, trunc, info = venv.step(acts)
        torch.cuda.synchronize()
        dt = time.time() - t
        print(f'[{name}] episode {ep}: {dt:.3f}s  {N * 288 / dt / 1e6:.1f} M env-steps/s')
    if name == 'device':
        eng = venv._engine
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(5):
            eng.generate_episodes(0, N, 1, 0)
        ev1.record(); torch.cuda.synchronize()
        print(f'[device] generate_kernel: {ev0.elapsed_time(ev1) / 5:.3f} ms per {N} episodes')
    venv.close()
