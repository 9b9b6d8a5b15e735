#!/bin/bash
# FETCH_SIZE calibration for the streaming kernel's read shapes (run on the GPU box from the repo root).
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/fetch_calib; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o calib -- $REPO/tools/probes/fetch_calib > $OUT/expected.txt 2> $OUT/err.txt
cd $REPO
python - <<'PY'
import pandas as pd, glob
exp = {l.split()[1]: int(l.split()[2]) for l in open('gpurun_out/fetch_calib/expected.txt') if l.startswith('expected')}
df = pd.read_csv(glob.glob('gpurun_out/fetch_calib/pmc/**/calib_counter_collection.csv', recursive=True)[0])
print('kernel                      expected MB   FETCH_SIZE KiB (mean of dispatches)   bytes / (FETCH_SIZE * 1024)')
for name, want in exp.items():
    rows = df[df['Kernel_Name'].str.startswith(name)]
    if name == 'read_x4_linear':
        rows = rows[rows['Counter_Value'] > 0.75 * rows['Counter_Value'].max()]      # the full 1 GiB passes only
    v = rows['Counter_Value'].mean()
    print(f'{name:<26} {want / 1e6:>10.1f}   {v:>14.1f}   {want / (v * 1024):>8.3f}')
PY
