#!/bin/bash
# A/B of kernel build variants (sustaingym_amd/variants/lib_*.so) x grid caps; prints kernel ms.
for lib in sustaingym_amd/variants/lib_*.so; do
  for cap in 2048 4096 16384; do
    for proj in "" "--no-project"; do
      r=$(SUSTAINGYM_AMD_LIB=$PWD/$lib EVC_GRID_CAP=$cap python bench.py --no-cpu-baseline --steps 288 --warmup 96 $proj 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['solver_kernel_ms'])")
      echo "$(basename $lib) cap=$cap proj='$proj' ms_per_step,kernel_ms,solver_ms = $r"
    done
  done
done
