#!/bin/bash
# Kernel trace + SQ counters (two --pmc passes, no tracing beside them) of the fused rollout kernel, per env-step;
# prints one JSON line and leaves gpurun_out/pmc_roll_<site>_<episodes>_<policy>/ (copy the line into profiles/).  Usage: tools/profile_rollout.sh site episodes policy
REPO=$(pwd); TAG=pmc_roll_$1_$2_$3; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -o r -- python $REPO/tools/one_rollout.py $1 $2 $3 > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/b -o r -- python $REPO/tools/one_rollout.py $1 $2 $3 > /dev/null 2> $OUT/b.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o r -- python $REPO/tools/one_rollout.py $1 $2 $3 > /dev/null 2> $OUT/t.err
cd $REPO
python - $TAG <<'PY'
import pandas as pd, glob, sys, json
tag = sys.argv[1]
res = {}
for part in 'ab':
    f = glob.glob(f'gpurun_out/{tag}/{part}/**/r_counter_collection.csv', recursive=True)
    if not f: continue
    df = pd.read_csv(f[0])
    df = df[df['Kernel_Name'].str.contains('rollout_kernel')]
    res.update((df.groupby('Counter_Name')['Counter_Value'].mean() / (65536 * 288)).round(3).to_dict())
    res['VGPR'] = int(df['VGPR_Count'].iloc[0]); res['scratch'] = int(df['Scratch_Size'].iloc[0]); res['LDS'] = int(df['LDS_Block_Size'].iloc[0])
f = glob.glob(f'gpurun_out/{tag}/t/**/r_kernel_stats.csv', recursive=True)
if f:
    df = pd.read_csv(f[0]); df = df[df['Name'].str.contains('rollout_kernel')]
    res['kernel_avg_ms'] = round(float(df['AverageNs'].iloc[0]) / 1e6, 3)
print(tag, json.dumps(res))
json.dump(res, open(f'gpurun_out/{tag}.json', 'w'), indent=1)
PY
