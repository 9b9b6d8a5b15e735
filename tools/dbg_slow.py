import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from sustaingym_amd.engine import StepEngine
from sustaingym_amd.network import caltech_acn
from sustaingym_amd._lib import SESSION_DTYPE
from sustaingym_amd.synthetic import synthetic_moer
net = caltech_acn(); n = 54; N = 8
cc = [i for i in range(n) if net.evse_kind[i] == 1]
sess = np.zeros((N, 16), SESSION_DTYPE); req = np.zeros((N, 16)); ns = np.zeros(N, np.int32)
for e in range(N):
    k = 4 + (e % 4)
    for j in range(k):
        sess[e, j] = (0, 200, 150, cc[j]); req[e, j] = 50.0
    ns[e] = k
eng = StepEngine(net, N, project_action=True, bank_slots=N, max_sessions=16, moer_days=1, debug_outputs=True)
eng.upload_moer(synthetic_moer(1)); eng.upload_episodes(ns, sess, req, np.zeros(N, np.int32)); eng.reset(host=True)
a = np.ones((N, n), np.float32)
for t in range(4):
    out = eng.step(a)
    print('step', t + 1, 'slow count', eng.last_slow_count(), 'pod sums', out['projected'][:, cc].sum(1) * 32)
