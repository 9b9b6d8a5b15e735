#!/usr/bin/env python
"""tools/probes/instance_period.py — step period of the bench workload for eight INSTANCES created one after the other in one process (each closed
before the next): is a slow run a property of the process, or of where one engine's buffers happened to land?  Prints the device addresses too."""
import json
import sys

sys.path.insert(0, '.')
import bench  # noqa: E402

rows = []
keep = []
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    w = bench.EvWorkload('caltech', 65536, 0, 0, project=True, pipeline=2)
    torch = w.torch
    w.run(600)
    w.eng.join(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); w.run(1500); w.eng.join(); e.record(); torch.cuda.synchronize()
    per = s.elapsed_time(e) / 1500 * 1e3
    out = w.out
    rows.append({'instance': i, 'us_per_step': round(per, 2), 'obs_ptr': hex(out['obs'].data_ptr()), 'ring0_ptr': hex(w.ring[0].data_ptr()),
                 'reward_ptr': hex(out['reward'].data_ptr()), 'term_ptr': hex(out['terminated'].data_ptr()), 'bd_ptr': hex(out['breakdown'].data_ptr()) if out.get('breakdown') is not None else None})
    print(json.dumps(rows[-1]), flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == 'keep':
        keep.append(w)          # later instances land elsewhere
    else:
        w.close()
