# Feasibility of VERDICT r2 item 2(c): the 65 536-environment step as two independent half-batches on two streams
# (one launch's tail under the other's body).  Prints us per 65 536 env-steps for: one engine; two half engines on two
# streams; two half engines on one stream.
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import torch
import bench

dev = torch.device('cuda', 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 576
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def timed(ws, steps):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        for w in ws:
            w.run(1)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e6


out = {}
w = bench.EvWorkload('caltech', 65536, 0, 0)
timed([w], 288)
out['one_engine'] = [round(timed([w], K), 2) for _ in range(3)]
w.close()
streams = [torch.cuda.Stream(dev) for _ in range(parts)]
ws = []
for i, s in enumerate(streams):
    with torch.cuda.stream(s):
        ws.append(bench.EvWorkload('caltech', 65536 // parts, 0, 0, seed_base=1000 + 17 * i))
        torch.cuda.synchronize(dev)
timed(ws, 288)
out[f'{parts}_engines_{parts}_streams'] = [round(timed(ws, K), 2) for _ in range(3)]
for x in ws:
    x.close()
ws = [bench.EvWorkload('caltech', 65536 // parts, 0, 0, seed_base=1000 + 17 * i) for i in range(parts)]
timed(ws, 288)
out[f'{parts}_engines_1_stream'] = [round(timed(ws, K), 2) for _ in range(3)]
print(json.dumps(out))
