import sys, time, os
sys.path.insert(0, '.')
import numpy as np
from sustaingym_amd.engine import StepEngine
from sustaingym_amd.network import caltech_acn
sys.path.insert(0, 'tests')
from helpers import make_workload
net = caltech_acn()
for N in (1, 64):
  for proj in (True, False):
    for dbg in (True, False):
        wl = make_workload(net, N, bank_slots=max(N, 4), seed=3, moer_days=2)
        eng = StepEngine(net, N, project_action=proj, autoreset=True, bank_slots=max(N, 4), max_sessions=wl['sessions'].shape[1], moer_days=2, debug_outputs=dbg)
        eng.upload_moer(wl['moer']); eng.upload_episodes(wl['n_sessions'], wl['sessions'], wl['requested'], wl['moer_day'])
        eng.reset(host=True)
        a = np.random.default_rng(0).random((N, 54), dtype=np.float32)
        res = []
        for mode in ('0', None):
            if mode is None: os.environ.pop('EVC_HOST_DIRECT_MAX_BYTES', None)
            else: os.environ['EVC_HOST_DIRECT_MAX_BYTES'] = mode
            for _ in range(50): eng.step(a)
            t0 = time.perf_counter()
            for _ in range(300): eng.step(a)
            res.append((time.perf_counter() - t0) / 300 * 1e6)
        print(f'N={N} project={proj} debug={dbg}: copies {res[0]:.1f} us, direct {res[1]:.1f} us')
        eng.close()
