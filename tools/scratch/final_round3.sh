#!/bin/bash
# what produced profiles/r3h_*: the GPU suite, tools/profile.sh, the driver's command
TAG=${1:-r3h}
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2>/dev/null
python - <<PY
import json
for f in ("gpurun_out/$TAG/bench_plain.json", "gpurun_out/${TAG}_bench_driver.json", "gpurun_out/$TAG/bench_traced.json"):
    d = json.load(open(f)); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["frac"], r.get("step_period_ms"), r.get("half_launch_ms"))
PY
