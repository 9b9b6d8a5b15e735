#!/bin/bash
# tools/unbundle.sh lib.so outdir: extracts every gfx950 code object of the library (one per translation unit: the
# .hip_fatbin section is a concatenation of clang offload bundles) into outdir/dev_<i>.co
LIB=$1; OUT=$2
mkdir -p "$OUT"
objcopy -O binary --only-section=.hip_fatbin "$LIB" "$OUT/fat.bin"
python3 - "$OUT" <<'PY'
import sys, subprocess
out = sys.argv[1]
data = open(out + '/fat.bin', 'rb').read()
magic = b'__CLANG_OFFLOAD_BUNDLE__'
starts = []
i = data.find(magic)
while i >= 0:
    starts.append(i)
    i = data.find(magic, i + 1)
for k, s in enumerate(starts):
    e = starts[k + 1] if k + 1 < len(starts) else len(data)
    open(f'{out}/bundle_{k}.bin', 'wb').write(data[s:e])
    subprocess.check_call(['/opt/rocm/lib/llvm/bin/clang-offload-bundler', '--unbundle', '--type=o', f'--input={out}/bundle_{k}.bin',
                           '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={out}/dev_{k}.co'])
PY
