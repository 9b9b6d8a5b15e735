# us per step of bench.py's workload with the engine's pipelined halves (evc_set_pipeline) on / off, same process.
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import torch
import bench

site = sys.argv[1] if len(sys.argv) > 1 else 'caltech'
episodes = sys.argv[2] if len(sys.argv) > 2 else 'synthetic'
K = int(sys.argv[3]) if len(sys.argv) > 3 else 576
dev = torch.device('cuda', 0)
w = bench.EvWorkload(site, 65536, 0, 0, episodes=episodes)
w.run(288)
out = {'site': site, 'episodes': episodes}
for mode in (1, 2, 1, 2):
    w.eng.set_pipeline(mode)
    w.run(64)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    w.run(K)
    issue = (time.perf_counter() - t0) / K * 1e6
    w.eng.join()
    torch.cuda.synchronize(dev)
    out.setdefault(f'pipeline{mode}_us', []).append(round((time.perf_counter() - t0) / K * 1e6, 2))
    out.setdefault(f'pipeline{mode}_issue_us', []).append(round(issue, 2))
out['pipelined_steps'] = w.eng.pipelined_steps(ordered=True)
print(json.dumps(out))
