# Interleaved A/B of grid caps (EVC_GRID_CAP is read at evc_create): min / median us per step.
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sustaingym_amd import _lib
from sustaingym_amd.engine import StepEngine
from sustaingym_amd.network import caltech_acn
from sustaingym_amd.synthetic import synthetic_episodes, synthetic_moer
net = caltech_acn(); N, n = 65536, 54
caps = [int(c) for c in sys.argv[1].split(',')]
ns, sess, req, day = synthetic_episodes(8192, n, seed=1000, stride=64, moer_days=32)
moer = synthetic_moer(32, seed=7)
g = torch.Generator(device='cuda'); g.manual_seed(1234)
ring = [torch.rand((N, n), device='cuda', generator=g) for _ in range(8)]
res = {}
for project in (True, False):
    engines = {}
    for cap in caps:
        os.environ['EVC_GRID_CAP'] = str(cap)
        eng = StepEngine(net, N, project_action=project, autoreset=True, bank_slots=8192, max_sessions=64, moer_days=32)
        eng.upload_moer(moer); eng.upload_episodes(ns, sess, req, day); eng.reset()
        step, out = eng.make_stepper()
        for i in range(288): step(ring[i % 8].data_ptr())
        engines[cap] = (eng, step)
    for rep in range(6):
        for cap in caps:
            eng, step = engines[cap]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(288): step(ring[i % 8].data_ptr())
            torch.cuda.synchronize()
            res.setdefault((project, cap), []).append((time.perf_counter() - t0) / 288 * 1e6)
    for cap in caps:
        v = res[(project, cap)]
        print(f'project={project} cap={cap}: min {min(v):.2f} median {np.median(v):.2f} us/step  all {[round(x,1) for x in v]}')
    for eng, _ in engines.values(): eng.close()
