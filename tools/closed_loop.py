#!/usr/bin/env python
"""bench.py's secondary.closed_loop_65536 on its own (GPU box): `python tools/closed_loop.py`."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps(bench.secondary_closed_loop(0, 'continuous'), indent=1))
