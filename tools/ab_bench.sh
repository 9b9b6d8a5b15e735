#!/bin/bash
# A/B inside ONE GPU call (run-to-run variation between boxes is large): quad vs wave kernel,
# each measured 3x interleaved; prints ms/step, kernel ms, solver ms.
for rep in 1 2 3; do
  for kern in quad wave; do
    for proj in "" "--no-project"; do
      r=$(EVC_KERNEL=$kern python bench.py --no-cpu-baseline --steps 288 --warmup 96 $proj 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['solver_kernel_ms'])")
      echo "rep$rep kernel=$kern proj='$proj' ms_per_step,kernel_ms,solver_ms = $r"
    done
  done
done
