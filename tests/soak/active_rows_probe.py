"""How many constraint rows are ACTIVE at the projection of a greedy / random action on GMM days (CPU, oracle only):
decides how many rows an in-register cone solver must handle before the general iteration is needed.
    python tests/soak/active_rows_probe.py [site] [policy] [envs]"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from sustaingym_amd.event_generation import GMMsTraceGenerator  # noqa: E402
from sustaingym_amd.network import site_str_to_site  # noqa: E402

site = sys.argv[1] if len(sys.argv) > 1 else 'jpl'
policy = sys.argv[2] if len(sys.argv) > 2 else 'greedy'
E = int(sys.argv[3]) if len(sys.argv) > 3 else 6
warnings.simplefilter('ignore')
net = site_str_to_site(site)
n, m = net.num_stations, len(net.magnitudes)
A = net.constraint_matrix * np.exp(1j * np.deg2rad(net.phase_angles))[None, :]
onet = ob.OracleNetwork(net)
hist = np.zeros(m + 1, np.int64)
moved_steps = 0
rng = np.random.default_rng(0)
gen = GMMsTraceGenerator(site, 'Summer 2019', seed=3)
for e in range(E):
    table = gen.get_event_table()
    moer = gen.get_moer()
    env = ob.OracleEnv(onet, 36, True)
    obs = env.reset(table.sessions, table.requested, moer)
    for t in range(288):
        a = (obs[:n] > 0).astype(np.float32) if policy == 'greedy' else rng.random(n, dtype=np.float32)
        obs, r = env.step(a)
        x = np.array(r.projected[:n])
        if np.any(np.abs(x - a) > 1e-9):
            cur = np.abs(A @ (x * 32))
            act = int(np.sum(cur >= net.magnitudes * (1 - 1e-7)))
            hist[act] += 1
            moved_steps += 1
print(site, policy, 'steps with a moved action:', moved_steps, 'of', E * 288)
print('active rows at the projection -> steps:', {k: int(v) for k, v in enumerate(hist) if v})
