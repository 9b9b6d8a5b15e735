REPO=$(pwd); OUT=$REPO/gpurun_out/r6_icache; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|SQ_INST_LEVEL|SQC_" | head -40 > $OUT/counters.txt
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  EVC_ROLLOUT_WAVES=2 rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -o r -- python $REPO/tools/one_rollout.py caltech gmm greedy > /dev/null 2> $OUT/$tag.err
done
cd $REPO
python - <<'PY'
import pandas as pd, glob
for f in glob.glob('gpurun_out/r6_icache/**/r_counter_collection.csv', recursive=True):
    df = pd.read_csv(f); df = df[df['Kernel_Name'].str.contains('rollout_kernel')]
    print(f.split('/')[2], (df.groupby('Counter_Name')['Counter_Value'].mean() / (65536 * 288)).round(3).to_dict())
PY
cat gpurun_out/r6_icache/counters.txt | head -30
tail -3 gpurun_out/r6_icache/*.err | head -20
