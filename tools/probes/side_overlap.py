#!/usr/bin/env python
"""tools/probes/side_overlap.py — do an engine's two side streams run concurrently?  For six engines created one after the other (all kept alive):
the step period of the bench workload, and how long two spin kernels (torch.cuda._sleep), one per side stream, take together relative to one."""
import json
import sys

sys.path.insert(0, '.')
import bench  # noqa: E402

keep = []
for i in range(6):
    w = bench.EvWorkload('caltech', 65536, 0, 0, project=True, pipeline=2)
    torch = w.torch
    w.run(600); w.eng.join(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); w.run(1500); w.eng.join(); e.record(); torch.cuda.synchronize()
    per = s.elapsed_time(e) / 1500 * 1e3
    halves = w.eng.pipeline_halves()
    streams = [st for _, st in halves]
    def spin(which):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for st in which:
            st.wait_event(a)
            with torch.cuda.stream(st):
                torch.cuda._sleep(400000)
        for st in which:
            torch.cuda.current_stream().wait_stream(st)
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b)
    one = min(spin(streams[:1]) for _ in range(3))
    two = min(spin(streams) for _ in range(3))
    print(json.dumps({'instance': i, 'us_per_step': round(per, 2), 'spin_one_ms': round(one, 3), 'spin_both_ms': round(two, 3), 'ratio': round(two / one, 2)}), flush=True)
    keep.append(w)
