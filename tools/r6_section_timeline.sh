# round 6: where one iteration's time goes — twelve stamps inside every wavefront's second quad, at 1 and 3 wavefronts per SIMD
# Needs sustaingym_amd/variants/lib_timeline3.so (tools/build_variant.sh timeline3 "-mllvm -disable-machine-licm -DEVC_TIMELINE=3").
export SUSTAINGYM_AMD_LIB=$PWD/sustaingym_amd/variants/lib_timeline3.so
for cap in 256 768; do echo "== EVC_GRID_CAP=$cap"; EVC_TIMELINE_MODE=3 EVC_GRID_CAP=$cap python tools/wg_timeline.py 1 2>/dev/null | grep -v amdgpu.ids; done
