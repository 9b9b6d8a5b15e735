#!/bin/bash
# kernel trace of one JPL GMM day (per-kernel totals): tools/scratch/trace_jpl.sh [site]
SITE=${1:-jpl}
REPO=$(pwd); OUT=$REPO/gpurun_out/trace_$SITE; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $REPO/bench.py --phase sync --episodes gmm --site $SITE --steps 288 --warmup 288 --no-secondary --no-cpu-baseline --kernel-timing-steps 1 > $OUT/bench.json 2> $OUT/err
cd $REPO
python - $OUT <<'PY'
import pandas as pd, glob, sys
f = glob.glob(sys.argv[1] + '/**/t_kernel_stats.csv', recursive=True)[0]
df = pd.read_csv(f)
print(df[['Name','Calls','TotalDurationNs','AverageNs','MaxNs']].head(8).to_string())
PY
