"""A longer run of tests/test_gpu_kkt_certificate.py's lock-step certification (GPU box; not collected by pytest):
32 768 environments per site from 05:00 to midnight of a GMM day, every step certified.  `python tests/soak/kkt_soak.py [caltech jpl]`"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_kkt_certificate as T  # noqa: E402
from test_gpu_rollout import _gmm_engine  # noqa: E402

N = int(os.environ.get('KKT_SOAK_N', 32768))
for site in (sys.argv[1:] or ['caltech', 'jpl']):
    period = 'Summer 2019' if site == 'caltech' else 'Summer 2021'
    t0 = time.time()
    net, dbg = _gmm_engine(site, period, N, N, seed=909, debug_outputs=True)
    _, lean = _gmm_engine(site, period, N, N, seed=909)
    dbg.set_tie_grid(40); lean.set_tie_grid(40)
    tally = T._Tally()
    T._lockstep(net, dbg, lean, tally, t0=60, t1=288, seed=77)
    tally.record(f'soak_{site}')
    tally.check(min_congested=1)
    print(site, json.dumps(T.REPORT[f'soak_{site}']), f'{time.time() - t0:.0f} s', flush=True)
    dbg.close(); lean.close()
