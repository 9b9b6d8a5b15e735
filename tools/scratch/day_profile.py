# Streaming-kernel time by time of day (per-launch events), bench workload; project on / off.
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from sustaingym_amd.engine import StepEngine
from sustaingym_amd.network import caltech_acn
from sustaingym_amd.synthetic import synthetic_episodes, synthetic_moer
net = caltech_acn(); N, n = 65536, 54
ns, sess, req, day = synthetic_episodes(8192, n, seed=1000, stride=64, moer_days=32)
moer = synthetic_moer(32, seed=7)
g = torch.Generator(device='cuda'); g.manual_seed(1234)
ring = [torch.rand((N, n), device='cuda', generator=g) for _ in range(8)]
zero = torch.zeros((N, n), device='cuda')
for project, acts in ((True, ring), (False, ring), (True, [zero])):
    eng = StepEngine(net, N, project_action=project, autoreset=True, bank_slots=8192, max_sessions=64, moer_days=32)
    eng.upload_moer(moer); eng.upload_episodes(ns, sess, req, day); eng.reset()
    step, out = eng.make_stepper()
    for i in range(288): step(acts[i % len(acts)].data_ptr())
    eng.enable_timing(True)
    ms = []
    for i in range(288):
        step(acts[i % len(acts)].data_ptr())
        ms.append(eng.last_step_ms()[0])
    ms = np.array(ms) * 1e3
    print(f'project={project} zero_actions={len(acts) == 1}: mean {ms.mean():.2f}; by 24-period bucket:', np.round(ms.reshape(12, 24).mean(1), 1))
    eng.close()
# occupancy of the workload by period
a = sess['arrival'].astype(int); d = sess['departure'].astype(int)
valid = np.arange(a.shape[1])[None, :] < ns[:, None]
occ = [(valid & (a <= t) & (d > t)).sum(1).mean() for t in range(0, 288, 24)]
arr = [(valid & (a >= t) & (a < t + 24)).sum(1).mean() / 24 for t in range(0, 288, 24)]
print('mean plugged-in EVs at bucket start:', np.round(occ, 1))
print('arrivals per env-step in bucket:', np.round(arr, 3))
