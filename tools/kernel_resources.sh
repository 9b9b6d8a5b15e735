#!/bin/bash
# tools/kernel_resources.sh [lib.so] [filter-regex]: VGPR / AGPR / SGPR / spills / LDS / scratch of every gfx950 kernel
# in the built library (reads the code object's metadata; no GPU needed).
LIB=${1:-sustaingym_amd/libevcharge_hip.so}
T=$(mktemp -d)
"$(dirname "$0")/unbundle.sh" "$LIB" $T
for co in $T/dev_*.co; do /opt/rocm/lib/llvm/bin/llvm-readelf --notes $co; done | python3 -c "
import re, subprocess, sys
txt = sys.stdin.read()
rows = []
for blk in txt.split('- .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    g = lambda k: int(re.search(r'\.' + k + r':\s+(\d+)', blk).group(1))
    ag = int(re.match(r':\s+(\d+)', blk).group(1))
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace('void evc::', '').replace('(evc::Params, evc::StepIO)', '').replace('(evc::Params, evc::RolloutIO)', '')
    rows.append((dem, g('vgpr_count'), ag, g('sgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))
flt = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
print(f'{\"kernel\":<52} vgpr agpr sgpr vspill sspill    lds scratch')
for r in sorted(rows):
    if flt is None or flt.search(r[0]):
        print(f'{r[0][:52]:<52} {r[1]:>4} {r[2]:>4} {r[3]:>4} {r[4]:>6} {r[5]:>6} {r[6]:>6} {r[7]:>7}')
" "${2:-}"
rm -rf $T
