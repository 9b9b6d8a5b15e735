# us per step of a window of K steps that starts from an idle, joined engine (what a short timed window sees)
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np, torch
import bench
dev = torch.device('cuda', 0)
w = bench.EvWorkload('caltech', 65536, 0, 0)
w.run(600)
out = {}
for mode in (1, 2):
    w.eng.set_pipeline(mode)
    for K in (5, 10, 20, 40, 80, 160, 320, 640):
        ts = []
        for rep in range(7):
            w.eng.join(); torch.cuda.synchronize(dev)
            time.sleep(0.002)
            t0 = time.perf_counter()
            w.run(K)
            torch.cuda.synchronize(dev)
            ts.append((time.perf_counter() - t0) * 1e6)
        out[f'p{mode}_K{K}'] = [round(float(np.median(ts)) / K, 2), round(float(np.median(ts)), 1)]
print(json.dumps(out))
