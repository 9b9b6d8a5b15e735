# Two engines, same inputs, 600 steps at N = 65536 on GMM days with projection: bitwise-identical outputs and state?
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sustaingym_amd.engine import StepEngine
from sustaingym_amd.network import site_str_to_site
from sustaingym_amd.event_generation import gmm_device_tables
from sustaingym_amd.synthetic import synthetic_moer
net = site_str_to_site('caltech'); N, n = 65536, 54
tabs = gmm_device_tables('caltech', 'Summer 2019')
moer = synthetic_moer(tabs['num_days'], seed=7)
def make():
    e = StepEngine(net, N, project_action=True, autoreset=True, bank_slots=8192, max_sessions=128, moer_days=tabs['num_days'])
    e.upload_moer(moer); e.upload_gmm(tabs); e.generate_episodes(0, 8192, 1, 0); e.reset()
    return e
a, b = make(), make()
g = torch.Generator(device='cuda'); g.manual_seed(1)
ring = [torch.rand((N, n), device='cuda', generator=g) for _ in range(8)]
bad = 0
for t in range(600):
    oa = a.step(ring[t % 8]); ob = b.step(ring[t % 8])
    if t % 25 == 0 or t > 590:
        for k in ('obs', 'reward', 'terminated', 'breakdown'):
            if not torch.equal(oa[k], ob[k]):
                bad += 1; print('MISMATCH', t, k)
sa, sb = a.get_state(), b.get_state()
for k in sa:
    if not np.array_equal(sa[k], sb[k]): bad += 1; print('STATE MISMATCH', k)
print('determinism check:', 'OK' if bad == 0 else f'{bad} mismatches', '| status census', int((a.env_scalars()['status'] != 0).sum()))
