#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=1 ..."  -> sustaingym_amd/variants/lib_NAME.so (A/B with tools/ab_libs.py)
set -e
mkdir -p sustaingym_amd/variants
cd sustaingym_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Wno-unused-function $2 -shared -o ../variants/lib_$1.so evc_engine.hip bat_engine.hip 2>&1 | grep -i "error" || true
