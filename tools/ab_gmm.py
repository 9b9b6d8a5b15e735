# us/step on a bank of device-generated GMM episodes (the reference's episode distribution) per period of the day
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sustaingym_amd.engine import StepEngine
from sustaingym_amd.network import site_str_to_site
from sustaingym_amd.event_generation import gmm_device_tables
from sustaingym_amd.synthetic import synthetic_moer
site = sys.argv[1] if len(sys.argv) > 1 else 'caltech'
net = site_str_to_site(site); N, n = 65536, net.num_stations
tabs = gmm_device_tables(site, 'Summer 2019')
g = torch.Generator(device='cuda'); g.manual_seed(1234)
ring = [torch.rand((N, n), device='cuda', generator=g) for _ in range(8)]
for project in (True, False):
    eng = StepEngine(net, N, project_action=project, autoreset=True, bank_slots=8192, max_sessions=128, moer_days=tabs['num_days'])
    eng.upload_moer(synthetic_moer(tabs['num_days'], seed=7)); eng.upload_gmm(tabs); eng.generate_episodes(0, 8192, 1, 0); eng.reset()
    step, out = eng.make_stepper()
    for i in range(288): step(ring[i % 8].data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    marks = []
    for i in range(288):
        step(ring[i % 8].data_ptr())
        if i % 48 == 47:
            torch.cuda.synchronize(); marks.append(time.perf_counter())
    per = np.diff([t0] + marks) / 48 * 1e6
    ns = eng.download_episodes(0, 8192, tables=False)[0]
    print(f'{site} gmm project={project}: {(marks[-1]-t0)/288*1e6:.1f} us/step; by 4h block {np.round(per,1)}; sessions/day mean {ns.mean():.1f} max {ns.max()}')
    eng.close()
