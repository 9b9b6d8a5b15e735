#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/trace_gmm; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $REPO/tools/ab_gmm.py ${1:-caltech} > $OUT/log 2> $OUT/err
cd $REPO; python - <<'PY'
import pandas as pd, glob, numpy as np
f = glob.glob('gpurun_out/trace_gmm/**/t_kernel_trace.csv', recursive=True)[0]
df = pd.read_csv(f)
df['dur'] = (df['End_Timestamp'] - df['Start_Timestamp']) / 1e3
for key in ('cquad<true', 'solver_step'):
    d = df[df['Kernel_Name'].str.contains(key, regex=False)]['dur'].values
    d = d[288:576]      # the timed day of the project=True run
    blocks = [d[i*48:(i+1)*48] for i in range(6)]
    print(key, 'us by 4h block: mean', [round(float(b.mean()),1) for b in blocks], 'max', [round(float(b.max()),1) for b in blocks])
PY
