# Fused rollout kernel (greedy / random device policies, warm-started projections under greedy) against the oracle's episode
# loop on device-generated GMM days, several seeds: counts instead of asserting.
#   python tests/soak/rollout_soak.py [site] [N] [seeds] [period]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import binding as ob
from sustaingym_amd.engine import StepEngine
from sustaingym_amd.event_generation import gmm_device_tables
from sustaingym_amd.hostio import to_host
from sustaingym_amd.network import site_str_to_site
from sustaingym_amd.synthetic import synthetic_moer
site = sys.argv[1] if len(sys.argv) > 1 else 'caltech'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
period = sys.argv[4] if len(sys.argv) > 4 else ('Summer 2019' if site == 'caltech' else 'Summer 2021')
net = site_str_to_site(site); n = net.num_stations
tabs = gmm_device_tables(site, period)
moer = synthetic_moer(tabs['num_days'], seed=3)
t0 = time.time()
for policy in ('greedy', 'random'):
    tot = dict(envs=0, ret_rel_gt_1e9=0, est=0, dep=0, rem_abs_gt_1e6=0, noconv=0)
    worst = 0.0
    for seed in range(seeds):
        eng = StepEngine(net, N, project_action=True, bank_slots=N, max_sessions=128, moer_days=tabs['num_days'])
        eng.upload_gmm(tabs); eng.upload_moer(moer); eng.generate_episodes(0, N, 500 + seed, 31 * seed)
        ns, sess, req, day, _ = eng.download_episodes(0, N)
        bat = ob.OracleBatch(ob.OracleNetwork(net), N, 36, True)
        bat.set_bank(ns, sess, req, day, moer)
        eng.set_policy_seed(77 + seed, env_id_base=1000 * seed)
        obs0 = to_host(eng.reset()).copy()
        assert np.array_equal(obs0, bat.reset())
        out = eng.rollout(policy=policy, steps=288)
        torch.cuda.synchronize()
        g = {k: to_host(v).copy() for k, v in out.items()}
        o = bat.rollout(policy, obs0, steps=288, seed=77 + seed, env_id_base=1000 * seed)
        rel = np.abs(g['returns'] - o['returns']) / np.maximum(1e-12, np.abs(o['returns']))
        worst = max(worst, float(rel.max()))
        rem, dep, est = eng.station_state(); orem, odep, oest = bat.station_state()
        tot['envs'] += N; tot['ret_rel_gt_1e9'] += int((rel > 1e-9).sum())
        tot['est'] += int((est != oest).sum()); tot['dep'] += int((dep != odep).sum())
        tot['rem_abs_gt_1e6'] += int((np.abs(rem - orem) > 1e-6).sum())
        tot['noconv'] += int(((eng.env_scalars()['status'] & 2) != 0).sum())
        eng.close()
    print(site, policy, tot, f'worst relative return difference {worst:.2e}', f'{time.time() - t0:.0f} s', flush=True)
