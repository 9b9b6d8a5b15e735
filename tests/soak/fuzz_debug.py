# Debug aid for tests/soak/network_fuzz.py: python tests/soak/fuzz_debug.py <seed> <case> <layout> <t> <env>
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
np.set_printoptions(precision=9, linewidth=220, suppress=True)
from helpers import make_pair, make_workload, random_network
seed, target, layout, T, E = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
rng = np.random.default_rng(seed)
for case in range(target + 1):
    net = random_network(rng, f'fuzz{case}')
n = net.num_stations; M = np.asarray(net.constraint_matrix); ph = np.asarray(net.phase_angles); mag = np.asarray(net.magnitudes)
print('n', n, 'm', len(mag), 'mags', mag, 'kind', np.asarray(net.evse_kind))
N = 64
wl = make_workload(net, N, seed=100 + target, busy=bool(target % 2), stride=96)
os.environ['EVC_LAYOUT'] = layout
eng, ob = make_pair(net, N, wl, project=True, debug=True)
eng.reset(host=True); ob.reset()
arng = np.random.default_rng(target)
for t in range(T + 1):
    a = arng.random((N, n), dtype=np.float32)
    if t % 50 == 25: a[::3] = 1.0
    if t == T: prev_obs = ob_last['obs'][E].copy() if t else None
    g = eng.step(a); o = ob.step(a); ob_last = o
print('t', T, 'env', E, 'slow count', eng.last_slow_count(), 'status', eng.env_scalars()['status'][E])
bad = np.flatnonzero(g['pilots'][E] != o['pilots'][E]); print('stations differing', bad)
sel = np.flatnonzero((g['projected'][E] > 0) | (o['projected'][E] > 0))
print('active stations', sel)
print('action*32 ', (a[E] * 32)[sel]); print('pilots g   ', g['pilots'][E][sel]); print('pilots o   ', o['pilots'][E][sel])
print('proj g*32  ', (g['projected'][E] * 32)[sel]); print('proj o*32  ', (o['projected'][E] * 32)[sel])
Z = M * np.exp(1j * np.deg2rad(ph))
for name, pr in (('g', g['projected'][E]), ('o', o['projected'][E])):
    print(name, 'row |.| - limit:', np.abs(Z @ (pr * 32)) - mag, ' dist to action', np.linalg.norm(pr - np.clip(a[E], 0, 1)))
print('classes of differing stations: rows touching them'); print(M[:, bad])
