# Kernel-level A/B leg (run through tools/ab_libs.py): mean begin-to-end time of the streaming kernel
# over whole days of the bench workload, from the engine's per-launch events (evc_enable_timing).
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from sustaingym_amd.engine import StepEngine
from sustaingym_amd.network import caltech_acn
from sustaingym_amd.synthetic import synthetic_episodes, synthetic_moer
net = caltech_acn(); N, n = 65536, 54
ns, sess, req, day = synthetic_episodes(8192, n, seed=1000, stride=64, moer_days=32)
moer = synthetic_moer(32, seed=7)
g = torch.Generator(device='cuda'); g.manual_seed(1234)
ring = [torch.rand((N, n), device='cuda', generator=g) for _ in range(8)]
for project in (True, False):
    eng = StepEngine(net, N, project_action=project, autoreset=True, bank_slots=8192, max_sessions=64, moer_days=32,
                     charge_calculation=os.environ.get('EVC_AB_BATTERY', 'continuous'))
    eng.upload_moer(moer); eng.upload_episodes(ns, sess, req, day); eng.reset()
    step, out = eng.make_stepper()
    for i in range(288): step(ring[i % 8].data_ptr())
    eng.enable_timing(True)
    ms = []
    for i in range(576):
        step(ring[i % 8].data_ptr())
        ms.append(eng.last_step_ms()[0])
    ms = np.array(ms) * 1e3
    print(f'project={project} kernel: min {ms.mean():.2f} us mean over 576 launches (night {ms[:288][200:].mean():.2f}, peak {ms.max():.2f})')
    eng.close()
