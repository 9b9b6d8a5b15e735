import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
os.environ['EVC_KERNEL'] = sys.argv[1] if len(sys.argv) > 1 else 'wave'
import numpy as np
from helpers import make_pair, make_workload
from sustaingym_amd.network import caltech_acn
net = caltech_acn(); n = 54; N = 64
wl = make_workload(net, N, seed=23 + N, busy=True)
eng, bat = make_pair(net, N, wl, project=True)
eng.reset(host=True); bat.reset()
rng = np.random.default_rng(N)
prev = None
for t in range(120):
    a = rng.random((N, n), dtype=np.float32) ** 0.5
    g = eng.step(a); o = bat.step(a)
    g = {k: v.copy() for k, v in g.items()}
    d = np.abs(g['projected'] - o['projected'])
    bad = d > 1e-12
    if bad.mean() > 1e-3:
        idx = np.argwhere(bad)
        print('step', t + 1, 'nbad', bad.sum(), 'max', d.max(), 'pilots equal', np.array_equal(g['pilots'], o['pilots']))
        for e, i in idx[:12]:
            print('  env', e, 'st', i, 'gpu', repr(g['projected'][e, i] * 32), 'orc', repr(o['projected'][e, i] * 32), 'a*32', a[e, i] * 32, 'diff_grid', (g['projected'][e, i] - o['projected'][e, i]) * 32 * 65536)
        A = (1/60)*(208/1000)*5
        for e in np.unique(idx[:, 0])[:3]:
            dem = prev['obs'][e, :n].astype(np.float64)
            b = a[e].astype(np.float64) * 32; h = np.minimum(32.0, dem / A)
            for pod in (range(10, 18), range(18, 26)):
                sel = list(pod)
                y0 = np.minimum(b, h)[sel]
                if y0.sum() <= 80: continue
                lo, hi = 0.0, 64.0
                for _ in range(200):
                    mid = 0.5 * (lo + hi)
                    if np.clip(b[sel] - mid, 0, h[sel]).sum() > 80: lo = mid
                    else: hi = mid
                yex = np.clip(b[sel] - 0.5 * (lo + hi), 0, h[sel])
                print('  env', e, 'pod', sel[0], 'exact', yex, '\n     gpu-exact', g['projected'][e, sel] * 32 - yex, '\n     orc-exact', o['projected'][e, sel] * 32 - yex)
            cur = np.abs(net.a_tilde() @ (o['projected'][e] * 32)); print('  rows orc', cur - net.magnitudes)
            cur = np.abs(net.a_tilde() @ (g['projected'][e] * 32)); print('  rows gpu', cur - net.magnitudes)
        envs = np.unique(idx[:, 0]); print('  envs', envs[:10], 'status', eng.env_scalars()['status'][envs[:5]])
        break
    prev = o
