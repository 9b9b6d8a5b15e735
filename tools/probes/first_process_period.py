#!/usr/bin/env python
"""tools/probes/first_process_period.py — step period of the bench workload in windows of 20 000 steps over the first ~8 s of GPU work of a
process (run it as the FIRST GPU process on a fresh box, then again): does a slow start go away with time under load?"""
import json
import sys
import time

sys.path.insert(0, '.')
import bench  # noqa: E402

w = bench.EvWorkload('caltech', 65536, 0, 0, project=True, pipeline=2)
torch = w.torch
out = []
t_start = time.time()
for i in range(22):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w.eng.join(); torch.cuda.synchronize()
    s.record()
    w.run(20000)
    w.eng.join()
    e.record()
    torch.cuda.synchronize()
    out.append(round(s.elapsed_time(e) / 20000 * 1e3, 2))
print(json.dumps({'us_per_step_by_20000_step_window': out, 'seconds': round(time.time() - t_start, 1)}))
w.close()
