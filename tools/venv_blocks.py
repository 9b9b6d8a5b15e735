"""EVChargingVectorEnv.step(output='torch') by 4-hour block of the day, pipeline=1 against pipeline=2 (GPU box):
where does the pipelined form through the API lose to the single launch?  `python tools/venv_blocks.py [N]`"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from sustaingym_amd.envs import EVChargingVectorEnv
from sustaingym_amd.event_generation import DeviceGMMTraceGenerator

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for pipe in (1, 2, 1, 2):
    venv = EVChargingVectorEnv(DeviceGMMTraceGenerator('caltech', 'Summer 2019', seed=0), num_envs=N, output='torch', pipeline=pipe)
    venv.reset(seed=0)
    acts = torch.rand((N, venv.num_stations), device='cuda')
    for _ in range(288):
        venv.step(acts)
    venv.join() if pipe == 2 else None
    torch.cuda.synchronize()
    rows = []
    t_all = time.perf_counter()
    for ep in range(2):
        marks = [time.perf_counter()]
        host = 0.0
        for s in range(288):
            h0 = time.perf_counter()
            venv.step(acts)
            host += time.perf_counter() - h0
            if s % 48 == 47:
                if pipe == 2: venv.join()
                torch.cuda.synchronize(); marks.append(time.perf_counter())
        rows.append((np.round(np.diff(marks) / 48 * 1e6, 1), round(host / 288 * 1e6, 1)))
    total = (time.perf_counter() - t_all) / 576 * 1e6
    print(f'pipeline={pipe}: {total:.1f} us/step (with a sync every 48 steps); by 4h block: {rows[0][0]} | {rows[1][0]}; host us/step {rows[0][1]} {rows[1][1]}; split steps {venv._engine.pipelined_steps()}')
    venv.close()
