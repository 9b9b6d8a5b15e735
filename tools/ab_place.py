# Does step time depend on where the buffers land?  Same config, several engine instances
# (fresh allocations each; padding allocations in between shift the addresses).
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sustaingym_amd.engine import StepEngine
from sustaingym_amd.network import caltech_acn
from sustaingym_amd.synthetic import synthetic_episodes, synthetic_moer
net = caltech_acn(); N, n = 65536, 54
ns, sess, req, day = synthetic_episodes(8192, n, seed=1000, stride=64, moer_days=32)
moer = synthetic_moer(32, seed=7)
pads = []
for inst in range(6):
    g = torch.Generator(device='cuda'); g.manual_seed(1234)
    ring = [torch.rand((N, n), device='cuda', generator=g) for _ in range(8)]
    eng = StepEngine(net, N, project_action=True, autoreset=True, bank_slots=8192, max_sessions=64, moer_days=32)
    eng.upload_moer(moer); eng.upload_episodes(ns, sess, req, day); eng.reset()
    step, out = eng.make_stepper()
    for i in range(288): step(ring[i % 8].data_ptr())
    ts = []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(288): step(ring[i % 8].data_ptr())
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 288 * 1e6)
    print(f'instance {inst}: min {min(ts):.2f} us/step; obs@{hex(out["obs"].data_ptr())} ring0@{hex(ring[0].data_ptr())}')
    pads.append(torch.empty((inst + 1) * 1234567, dtype=torch.uint8, device='cuda'))   # shift later allocations
    pads.append(eng); pads.append(ring)
