#!/bin/bash
# A/B of library variants inside ONE GPU call, interleaved twice.
for rep in 1 2; do
for lib in sustaingym_amd/variants/lib_*.so; do
  for proj in "" "--no-project"; do
      r=$(SUSTAINGYM_AMD_LIB=$PWD/$lib python bench.py --no-cpu-baseline --steps 288 --warmup 96 $proj 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['solver_kernel_ms'])")
      echo "rep$rep $(basename $lib) proj='$proj' ms_per_step,kernel_ms,solver_ms = $r"
  done
done
done
