# BASELINE config 5: MultiAgentEVChargingEnv, 8192 envs x 54 agents, per-agent observation gather.
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sustaingym_amd.envs import MultiAgentEVChargingVectorEnv
from sustaingym_amd.event_generation import DeviceGMMTraceGenerator
N = 8192
for name, kw in (('zero-copy view (reference semantics: every agent sees the same array)', dict()),
                 ('materialised [N, n, F] per-agent rows', dict(materialize=True)),
                 ('documented periods_delay=2 (own current + others delayed)', dict(periods_delay=2, delay_semantics='documented'))):
    env = MultiAgentEVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2021', seed=1), num_envs=N, **kw)
    obs = env.reset(seed=0)
    acts = torch.rand((N, env.num_agents), device='cuda')
    for _ in range(20): env.step(acts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps = 200
    for _ in range(steps): out = env.step(acts)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    ob = out[0]
    nbytes = ob.numel() * 4 if kw else 0
    print(f'{name}: {dt*1e6:.1f} us/step, {N*env.num_agents/dt/1e9:.2f} G agent-steps/s, obs {tuple(ob.shape)}' + (f', gather {nbytes/dt/1e12:.2f} TB/s written' if nbytes else ''))
    env.close()
