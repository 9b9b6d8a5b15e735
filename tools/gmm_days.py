#!/usr/bin/env python
"""bench.py's secondary.gmm_caltech / gmm_jpl records on their own (GPU box): `python tools/gmm_days.py [caltech jpl]`.
With SUSTAINGYM_AMD_LIB pointing at a variant library this is the A/B of a kernel change on the congested days."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

for site in (sys.argv[1:] or ['caltech', 'jpl']):
    r = bench.secondary_days(site, 'gmm', 0, 'continuous')
    print(json.dumps({k: r[k] for k in ('ms_per_step', 'env_steps_per_s', 'single_launch', 'kernel_us_by_4h', 'solver_kernel_us_by_4h',
                                        'slow_queue_envs_per_step')} | {'site': site}))
