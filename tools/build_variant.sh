#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=1 ..." [tu]  -> sustaingym_amd/variants/lib_NAME.so (A/B with tools/ab_libs.py).
# Recompiles ONE translation unit with the flags (tu = evc_engine (default) | evc_rollout | bat_engine) and links it
# with the objects of the regular build (run `make -C sustaingym_amd/csrc -j4` first).
set -e
TU=${3:-evc_engine}
mkdir -p sustaingym_amd/variants
cd sustaingym_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Wno-unused-function $2 -c -o /tmp/variant_$1_$TU.o $TU.hip 2>&1 | grep -i "error" || true
OBJS=""
for o in evc_engine evc_rollout bat_engine; do
  if [ $o = $TU ]; then OBJS="$OBJS /tmp/variant_$1_$TU.o"; else OBJS="$OBJS $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/lib_$1.so $OBJS
