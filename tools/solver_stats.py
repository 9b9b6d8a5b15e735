import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
os.environ.setdefault('SUSTAINGYM_AMD_LIB', os.path.join(os.getcwd(), 'sustaingym_amd/variants/lib_stats.so'))
import numpy as np, torch
from sustaingym_amd import _lib
from sustaingym_amd.engine import StepEngine
from sustaingym_amd.network import site_str_to_site
from sustaingym_amd.event_generation import gmm_device_tables
from sustaingym_amd.synthetic import synthetic_moer
site = sys.argv[1] if len(sys.argv) > 1 else 'caltech'
net = site_str_to_site(site); N, n = 65536, net.num_stations
tabs = gmm_device_tables(site, 'Summer 2019')
g = torch.Generator(device='cuda'); g.manual_seed(1234)
ring = [torch.rand((N, n), device='cuda', generator=g) for _ in range(8)]
eng = StepEngine(net, N, project_action=True, autoreset=True, bank_slots=8192, max_sessions=128, moer_days=tabs['num_days'])
eng.upload_moer(synthetic_moer(tabs['num_days'], seed=7)); eng.upload_gmm(tabs); eng.generate_episodes(0, 8192, 1, 0); eng.reset()
step, out = eng.make_stepper()
lib = _lib.load(); st = (C.c_ulonglong * 32)()
lib.evc_debug_solver_stats(st)
for blk in range(6):
    for i in range(48): step(ring[i % 8].data_ptr())
    lib.evc_debug_solver_stats(st)
    v = np.array(list(st), dtype=float)
    envs = max(v[0], 1); hard = max(v[0] - v[1], 1)
    if v[0] > 0: print('   raw', [int(x) for x in v])
    if v[0] > 0: print(f'   front per env: exact rows at the box clip {v[15]/envs:.0f}  caps filling + rows {v[16]/envs:.0f}  caps+worst-row cone (b2) {v[17]/envs:.0f} (entered {v[18]:.0f}, settled {v[19]:.0f}, caps/entry {v[20]/max(v[18],1):.2f})  capped cone (b0) {v[21]/max(v[22],1):.0f} per entry (entered {v[22]:.0f}, settled {v[23]:.0f})')
    if v[0] > 0: print(f'   cycles/env (100 MHz clock64 ticks): load+exact {v[8]/envs:.0f}  solve {v[9]/envs:.0f}  [head {v[10]/envs:.0f} build {v[11]/envs:.0f} chol {v[12]/envs:.0f} linesearch {v[13]/envs:.0f}]  finish {v[14]/envs:.0f}')
    print(f'{site} block {blk}: queued/step {v[0]/48:.0f}  settled w/o Newton {v[1]/envs:.2f}  iters/hard {v[2]/hard:.1f}  trials/hard {v[3]/hard:.1f}  active rows {v[4]/hard:.2f}  >=20 iters {v[5]/hard:.3f}  single-row {v[6]/hard:.2f}  noconv {v[7]:.0f}')
