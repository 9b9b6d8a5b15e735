# The bench.py workload itself (N = 65 536, 8192-episode bank, autoreset, projection on) replayed by the
# C oracle: every output of every step compared for two days.
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
import numpy as np, torch
from oracle.binding import OracleBatch, OracleNetwork
from sustaingym_amd.engine import StepEngine
from sustaingym_amd.network import caltech_acn
from sustaingym_amd.synthetic import synthetic_episodes, synthetic_moer
net = caltech_acn(); N, n, P = 65536, 54, 8192
ns, sess, req, day = synthetic_episodes(P, n, seed=1000, stride=64, moer_days=32)
moer = synthetic_moer(32, seed=7)
eng = StepEngine(net, N, project_action=True, autoreset=True, bank_slots=P, max_sessions=64, moer_days=32)
eng.upload_moer(moer); eng.upload_episodes(ns, sess, req, day); eng.set_autoreset_stride(1)
ob = OracleBatch(OracleNetwork(net), N, 36, True)
ob.set_bank(ns, sess, req, day, moer, autoreset_stride=1)
slots = (np.arange(N) % P).astype(np.int32)
assert np.array_equal(eng.reset(slots=slots, host=True), ob.reset(slots))
g = torch.Generator(device='cuda'); g.manual_seed(1234)
ring = [torch.rand((N, n), device='cuda', generator=g) for _ in range(8)]
ring_h = [r.cpu().numpy() for r in ring]
bad = dict(term=0, est=0, demand=0, moer=0, reward=0, final=0); worst = 0.0; t0 = time.time()
for t in range(576):
    go = {k: v.cpu().numpy() for k, v in eng.step(ring[t % 8]).items()}
    oo = ob.step(ring_h[t % 8], autoreset=True, debug=False)
    bad['term'] += int((go['terminated'] != oo['terminated']).sum())
    bad['est'] += int((go['obs'][:, n:2*n] != oo['obs'][:, n:2*n]).sum())
    bad['moer'] += int((go['obs'][:, 2*n:] != oo['obs'][:, 2*n:]).sum())
    d = np.abs(go['obs'][:, :n] - oo['obs'][:, :n]) / np.maximum(np.abs(oo['obs'][:, :n]), 1e-3)
    bad['demand'] += int((d > 1e-6).sum())
    r = np.abs(go['reward'] - oo['reward']) / np.maximum(np.abs(oo['reward']), 1e-3)
    bad['reward'] += int((r > 1e-9).sum()); worst = max(worst, float(r.max()))
    if oo['terminated'].any():
        m = oo['terminated'].astype(bool)
        bad['final'] += int((np.abs(go['final_obs'][m] - oo['final_obs'][m]) > 1e-5).sum())
met = eng.read_metrics()
print('bench workload, 576 steps x 65536 envs:', bad, f'worst relative reward difference {worst:.2e}',
      'episodes finished', int(met['episodes_finished']), 'envs with status', int(met['envs_with_status']),
      f'{time.time() - t0:.0f} s')
