# round 6: per-quad time of a wavefront of the streaming kernel at 1, 2 and 3 wavefronts per SIMD (grid caps 256 / 512 / 768 workgroups of
# four wavefronts; one launch per step) — is the iteration bound by its own latency chain or by what the wavefronts share?
# Needs sustaingym_amd/variants/lib_timeline.so (tools/build_variant.sh timeline "-mllvm -disable-machine-licm -DEVC_TIMELINE=1").
export SUSTAINGYM_AMD_LIB=$PWD/sustaingym_amd/variants/lib_timeline.so
for cap in 256 512 768; do echo "== EVC_GRID_CAP=$cap"; EVC_GRID_CAP=$cap python tools/wg_timeline.py 1 2>/dev/null | grep -v amdgpu.ids; done
