#!/usr/bin/env python
"""tools/probes/host_issue_profile.py — host time of each of the first 24 step() calls behind a device synchronisation (the driver's 20-step
window begins like this), five windows; pipelined halves and one launch per step."""
import json
import sys
import time

sys.path.insert(0, '.')
import bench  # noqa: E402

for pipeline in (2, 1):
    w = bench.EvWorkload('caltech', 65536, 0, 0, project=True, pipeline=pipeline)
    torch = w.torch
    w.run(600)
    rows = []
    for rep in range(5):
        w.run(5)
        w.eng.join(); torch.cuda.synchronize()
        ts = []
        t00 = time.perf_counter_ns()
        for i in range(24):
            t0 = time.perf_counter_ns()
            w.run(1)
            ts.append(round((time.perf_counter_ns() - t0) / 1e3, 1))
        t_issue = (time.perf_counter_ns() - t00) / 1e3
        w.eng.join(); torch.cuda.synchronize()
        t_all = (time.perf_counter_ns() - t00) / 1e3
        rows.append({'us_per_call': ts, 'issue_us': round(t_issue, 1), 'window_us': round(t_all, 1)})
    print(json.dumps({'pipeline': pipeline, 'windows': rows}))
    w.close()
