# per-step queue length and kernel durations over one GMM day: python tools/scratch/jpl_steps.py [site]
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, warnings
warnings.simplefilter('ignore')
from sustaingym_amd.engine import StepEngine
from sustaingym_amd.network import site_str_to_site
from sustaingym_amd.event_generation import gmm_device_tables
from sustaingym_amd.synthetic import synthetic_moer
site = sys.argv[1] if len(sys.argv) > 1 else 'jpl'
net = site_str_to_site(site); N, n = 65536, net.num_stations
tabs = gmm_device_tables(site, 'Summer 2019')
g = torch.Generator(device='cuda'); g.manual_seed(1234)
ring = [torch.rand((N, n), device='cuda', generator=g) for _ in range(8)]
eng = StepEngine(net, N, project_action=True, autoreset=True, bank_slots=8192, max_sessions=128, moer_days=tabs['num_days'])
eng.upload_moer(synthetic_moer(tabs['num_days'], seed=7)); eng.upload_gmm(tabs); eng.generate_episodes(0, 8192, 1, 0); eng.reset()
step, out = eng.make_stepper()
eng.enable_timing(True)
rows = []
prev = eng.read_metrics()['tie_snap_near_boundary'] if os.environ.get('UNDEC') else 0
for i in range(288):
    step(ring[i % 8].data_ptr())
    a, b = eng.last_step_ms()
    und = 0
    if os.environ.get('UNDEC'):
        cur = eng.read_metrics()['tie_snap_near_boundary']; und = cur - prev; prev = cur
    rows.append((i, eng.last_slow_count(), a * 1e3, b * 1e3, und))
r = np.array(rows)
for i in range(0, 288, 6):
    blk = r[i:i + 6]
    print(f'step {i:3d}: queue max {int(blk[:,1].max()):5d}  main {blk[:,2].mean():6.1f}  solver mean {blk[:,3].mean():6.1f} max {blk[:,3].max():6.1f}  undecided/step {blk[:,4].mean():8.0f}')
