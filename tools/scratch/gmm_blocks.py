# a synchronised GMM day by 4-hour block, one launch per step vs pipelined halves (host clock per block of 48 steps)
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np, torch
import bench
site = sys.argv[1] if len(sys.argv) > 1 else 'caltech'
dev = torch.device('cuda', 0)
w = bench.EvWorkload(site, 65536, 0, 0, episodes='gmm', phase='sync')
w.run(288)
out = {'site': site}
for mode in (1, 2, 1, 2):
    w.eng.set_pipeline(mode)
    blocks = []
    for b in range(6):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        w.run(48)
        torch.cuda.synchronize(dev)
        blocks.append(round((time.perf_counter() - t0) / 48 * 1e6, 1))
    out.setdefault(f'p{mode}', []).append(blocks + [round(float(np.mean(blocks)), 2)])
print(json.dumps(out))
