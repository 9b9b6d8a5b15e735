#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/trace_tmp; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $REPO/bench.py --steps 288 --warmup 96 --no-cpu-baseline "$@" > /dev/null 2> $OUT/err
cd $REPO; python - <<'PY'
import pandas as pd, glob
f = glob.glob('gpurun_out/trace_tmp/**/bench_kernel_stats.csv', recursive=True)[0]
df = pd.read_csv(f); df['Name'] = df['Name'].str.slice(0, 70)
print(df.head(6)[['Name','Calls','AverageNs','MinNs','MaxNs','Percentage']].to_string())
PY
